"""The C-ABI library loads and exports every symbol include/almpc.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "almpc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(almpc_[a-z_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    names = _declared()
    for must in ("almpc_create", "almpc_destroy", "almpc_design_shared", "almpc_set_reference",
                 "almpc_update_initialization", "almpc_calculate", "almpc_get_results", "almpc_get_design",
                 "almpc_last_error", "almpc_default_opts"):
        assert must in names


def test_library_exports_every_declared_symbol(capi):
    lib = ctypes.CDLL(capi.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, f"libalmpc.so does not export {missing}"


def test_library_has_gfx950_code_object(capi):
    blob = open(capi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob, "no gfx950 code object embedded: the HIP kernels were not compiled for MI355X"
    assert b"k_admm" in blob and b"k_polish" in blob and b"k_rollout" in blob


def test_default_opts_are_osqp_defaults_plus_documented_changes(capi):
    o = capi.default_opts()
    assert (o.rho, o.sigma, o.alpha, o.eps_abs, o.eps_rel, o.check_every) == (0.1, 1e-6, 1.6, 1e-3, 1e-3, 25)
    assert o.polish == 1 and o.max_iter == 25 and o.warm_start == 0
    with pytest.raises(TypeError):
        capi.default_opts(no_such_option=1)


def test_opts_struct_layout_matches_header(capi):
    # 5 doubles + 8 int32 = 72 bytes, no padding surprises between C and ctypes
    assert ctypes.sizeof(capi.almpc_opts) == 72
    assert capi.almpc_opts.max_iter.offset == 40 and capi.almpc_opts.warm_start.offset == 56


def test_no_cpu_fallback_without_device(capi):
    """On a box without a GPU the product must fail loudly (ALMPC_ERR_NO_DEVICE), never compute on the CPU."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.AlmpcError) as ei:
        capi.Solver(2, 1, 10, 1)
    assert ei.value.code == -2


def test_create_rejects_bad_sizes(capi):
    lib = capi.load()
    h = ctypes.c_void_p()
    assert lib.almpc_create(ctypes.byref(h), 0, 1, 1, 1, 0, 0) == -1
    assert lib.almpc_create(ctypes.byref(h), 2, 1, 200, 1, 0, 0) == -4   # m*N > 128
    assert lib.almpc_create(ctypes.byref(h), 65, 1, 10, 1, 0, 0) == -4   # n > 64
    assert lib.almpc_create(None, 2, 1, 10, 1, 0, 0) == -1
    assert lib.almpc_calculate(None, None) == -1
    assert lib.almpc_last_error(None) == b"null handle"


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing in the package may import, include, link or load it
    (doc comments may cite it)."""
    import re
    pkg = os.path.join(ROOT, "automationlabsmodelpredictivecontrol.jl_amd")
    bad = re.compile(r"^\s*(from|import)\s+(mpc_oracle|c_oracle)|#\s*include[^\n]*oracle|dlopen\([^)]*oracle|libalmpc_oracle|CDLL\([^)]*oracle", re.M)
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert not bad.search(txt), fn
    mk = open(os.path.join(ROOT, "Makefile")).read()
    lib_rule = mk[mk.index("$(LIB):"):mk.index("$(ORACLE):")]
    assert "oracle" not in lib_rule


def test_dare_is_host_math_and_rejects_what_has_no_stabilising_solution(capi, mo):
    """almpc_dare (design-time host code, what ControlSystems.are does at src/sub/design_mpc.jl:327): agrees with scipy on the
    reference-held QTP model and on the benchmark plant; a model whose unstable mode is uncontrollable has no stabilising
    solution and must be an error, not a huge 'converged' P (the doubling iteration diverges there)."""
    import numpy as np
    import scipy.linalg as sla
    p = mo.quadrotor()
    P = capi.dare(p.A, p.B, p.Q, p.R)
    ref = sla.solve_discrete_are(p.A, p.B, p.Q, p.R)
    assert np.abs(P - ref).max() <= 1e-9 * np.abs(ref).max()
    for A, B in ((np.array([[1.2, 0.0], [0.0, 0.5]]), np.array([[0.0], [1.0]])),     # unstable and uncontrollable
                 (np.array([[1.0]]), np.array([[0.0]]))):                              # marginal and uncontrollable
        with pytest.raises(capi.AlmpcError) as ei:
            capi.dare(A, B, np.eye(len(A)), np.eye(1))
        assert ei.value.code == -6
