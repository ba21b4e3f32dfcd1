"""Host-side mirror of the reference interface: argument handling and error behaviour (CPU only)."""
import os

import numpy as np
import pytest

from conftest import ROOT


def test_design_reference_broadcast(pkg):
    # src/main/main_mpc.jl:105-117: x = x_ref .* ones(n, N+1), u = u_ref .* ones(m, N)
    r = pkg._design_reference_mpc([0.65] * 4, [1.2, 1.3], 5)
    assert r.x.shape == (4, 6) and r.u.shape == (2, 5)
    assert np.all(r.x == 0.65) and np.all(r.u[0] == 1.2) and np.all(r.u[1] == 1.3)


def test_weights_defaults_and_kws_idiom(pkg):
    sys_ = pkg.ConstrainedLinearControlDiscreteSystem(np.eye(3), np.ones((3, 2)), pkg.Hyperrectangle([0] * 3, [1] * 3),
                                                      pkg.Hyperrectangle([0, 0], [1, 1]))
    w = pkg._create_weights_coefficients(sys_)
    assert np.array_equal(w.Q, 100 * np.eye(3)) and np.array_equal(w.R, 0.1 * np.eye(2)) and not w.S.any()
    w = pkg._create_weights_coefficients(sys_, kws=dict(mpc_Q=7.0, mpc_S=2.0))  # kws passed through, as the reference does
    assert w.Q[0, 0] == 7.0 and w.S[1, 1] == 2.0
    w = pkg._create_weights_coefficients(sys_, mpc_R=3.0)
    assert w.R[0, 0] == 3.0


def _sys(pkg):
    return pkg.ConstrainedLinearControlDiscreteSystem([[1, 1], [0, 1]], [[0.5], [1.0]], pkg.Hyperrectangle([-5, -5], [5, 5]),
                                                      pkg.Hyperrectangle([-1], [1]))


def test_proceed_controller_argument_errors(pkg):
    s = _sys(pkg)
    assert pkg.proceed_controller(s, "economic_model_predictive_control", 10, 1, [0, 0], [0]) is None  # removed in v0.1.4
    with pytest.raises(TypeError):
        pkg.proceed_controller(s, "model_predictive_control", 10, 1.5, [0, 0], [0])  # sample time must be Int
    with pytest.raises(KeyError):
        pkg.proceed_controller(s, "model_predictive_control", 10, 1, [0, 0], [0], mpc_solver="highs")
    with pytest.raises(KeyError):
        pkg.proceed_controller(s, "model_predictive_control", 10, 1, [0, 0], [0], mpc_programming_type="quadratic")
    for kw in (dict(mpc_solver="osqp"), dict(mpc_programming_type="non_linear"), dict(mpc_terminal_ingredient="contractive")):
        with pytest.raises(NotImplementedError):
            pkg.proceed_controller(s, "model_predictive_control", 10, 1, [0, 0], [0], **kw)


def test_design_fails_loudly_without_gpu(pkg):
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg._capi.AlmpcError):
        pkg.proceed_controller(_sys(pkg), "model_predictive_control", 10, 1, [0, 0], [0])


def test_system_validation(pkg):
    with pytest.raises(ValueError):
        pkg.Hyperrectangle([1.0], [0.0])
    with pytest.raises(ValueError):
        pkg.ConstrainedLinearControlDiscreteSystem(np.eye(2), np.ones((3, 1)), pkg.Hyperrectangle([0, 0], [1, 1]),
                                                   pkg.Hyperrectangle([0], [1]))


@pytest.mark.parametrize("batch,world", [(4096, 1), (32768, 8), (10, 3), (5, 8), (0, 2)])
def test_shard_range_partitions(pkg, batch, world):
    rs = [pkg.shard_range(batch, r, world) for r in range(world)]
    assert rs[0][0] == 0 and rs[-1][1] == batch
    assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
    sizes = [b - a for a, b in rs]
    assert max(sizes) - min(sizes) <= 1 and sorted(sizes, reverse=True) == sizes
    with pytest.raises(ValueError):
        pkg.shard_range(batch, world, world)


def test_aggregate_rate(pkg):
    assert pkg.sharding.aggregate_rate(4096, 10, 2.0, 8) == 8 * 4096 * 10 / 2.0


def test_workloads_module_matches_the_oracle_generators(pkg, mo):
    """bench.py takes its synthetic inputs from the product-side workloads module (it may not route inputs through oracle/);
    the oracle keeps its own copies for the tests: the two must produce identical arrays."""
    import importlib
    wl = importlib.import_module(pkg.__name__ + ".workloads")
    assert np.array_equal(wl.splitmix_normal(0x5EED0002, 5, 9, 12), mo.splitmix_normal(0x5EED0002, 5, 9, 12))
    assert np.array_equal(wl.quadrotor_x0_batch(7, 3.0, first_instance=11), mo.quadrotor_x0_batch(7, 3.0, first_instance=11))
    A, B = wl.quadrotor_model()
    Ao, Bo = mo.quadrotor_model()
    assert np.array_equal(A, Ao) and np.array_equal(B, Bo)
    q, qo = wl.quadrotor(30), mo.quadrotor(30)
    for k in ("Q", "R", "S", "u_min", "u_max", "x_ref", "u_ref"):
        assert np.array_equal(getattr(q, k), getattr(qo, k)), k
    W_in, W_h, b_h, W_out = wl.synthetic_fnn_weights()
    fo = mo.synthetic_fnn(act="tanh")
    A0, _ = mo.FnnModel(W_in, W_h, b_h, W_out, "tanh").jacobian(np.zeros(4), np.zeros(2))
    assert np.array_equal(W_in, fo.W_in) and np.array_equal(W_h[1], fo.W_h[1]) and np.array_equal(b_h[0], fo.b_h[0])
    assert np.array_equal(wl.scale_to_radius(W_out, A0), fo.W_out)
    # the reference's own test size: the product-side constants are the values the oracle decodes from the reference-held fixture
    import os
    from conftest import ROOT
    Af, Bf = mo.decode_linear_regressor_fixture(open(os.path.join(ROOT, "tests", "golden", "linear_regressor_train_result.jls"), "rb").read())
    qf, qfo = wl.qtp_fixture(), mo.qtp_linear_fixture_problem(Af, Bf)
    assert np.array_equal(qf.A, Af) and np.array_equal(qf.B, Bf)
    for k in ("Q", "R", "S", "u_min", "u_max", "x_ref", "u_ref"):
        assert np.array_equal(getattr(qf, k), getattr(qfo, k)), k


def test_reference_side_patch_applies_to_the_reference(tmp_path):
    """julia/reference_hip.patch (the reference-side edits of INTEGRATION.md as an actual diff: hip_solver_def tag, "hip" in
    _IMPLEMENTATION_SOLVER_LIST, the design branches of the linear-system AND the black-box method to
    _model_predictive_control_design_hip, batch-sized result arrays, the dispatch in update_initialization! / calculate!, and the
    shim itself as src/hip/AlmpcHIP.jl) applies cleanly to the reference's own files and is what julia/make_reference_patch.py
    produces from the current shim.  Build container only: the GPU box has no /root/reference."""
    import importlib.util
    import shutil
    import subprocess
    ref = "/root/reference"
    if not os.path.isdir(ref):
        pytest.skip("no reference checkout here")
    files = ["src/types/types.jl", "src/sub/solver_selection.jl", "src/sub/design_mpc.jl", "src/main/computation_mpc.jl",
             "src/AutomationLabsModelPredictiveControl.jl"]
    for f in files:
        os.makedirs(os.path.dirname(tmp_path / f), exist_ok=True)
        shutil.copy(os.path.join(ref, f), tmp_path / f)
    patch = os.path.join(ROOT, "julia", "reference_hip.patch")
    touched = {ln.split(" b/")[1].strip() for ln in open(patch) if ln.startswith("diff --git")}
    assert touched == set(files) | {"src/hip/AlmpcHIP.jl"}
    r = subprocess.run(["git", "apply", "--check", "--verbose", patch], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    subprocess.check_call(["git", "apply", patch], cwd=tmp_path)
    sel = open(tmp_path / "src/sub/solver_selection.jl").read()
    assert "hip = hip_solver_def()," in sel
    design = open(tmp_path / "src/sub/design_mpc.jl").read()
    assert "AlmpcHIP.design_hip(" in design
    # both design methods branch to the GPU design: the linear-system one and the black-box one
    assert design.count("if mpc_solver isa hip_solver_def") == 2
    assert "AlmpcHIP.design_relin_fnn!(" in design and "AlmpcHIP.design_sqp_fnn!(" in design and "AlmpcHIP.fnn_linearize(" in design
    # results of a batched controller are allocated with the batch (a single-instance allocation would be overrun by the library)
    assert "_memory_allocation_initialization_results_hip(" in design and "(horizon + 1) * batch" in design
    # the shim travels with the patch, identical to julia/AlmpcHIP.jl
    shim = open(os.path.join(ROOT, "julia", "AlmpcHIP.jl")).read()
    assert open(tmp_path / "src/hip/AlmpcHIP.jl").read() == shim
    spec = importlib.util.spec_from_file_location("make_reference_patch", os.path.join(ROOT, "julia", "make_reference_patch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.build(ref) == open(patch).read(), "julia/reference_hip.patch is stale: run python julia/make_reference_patch.py"
    # every ccall in the shim names a symbol the header declares
    import re
    hdr = open(os.path.join(ROOT, "include", "almpc.h")).read()
    for sym in set(re.findall(r"ccall\(\(:(almpc_[a-z_]+), libalmpc\)", shim)):
        assert re.search(r"\b" + sym + r"\s*\(", hdr), sym
    # the entry points that write through caller-supplied result pointers check the lengths first (a single-instance allocation with
    # mpc_batch > 1 must raise DimensionMismatch, not overrun the Julia heap)
    for fn in ("function calculate!(", "function read_results!(", "function results_wait!(", "function _model_predictive_control_computation(mod::HipModeler"):
        body = shim[shim.index(fn):]
        body = body[:body.index("\nend\n")]
        assert "check_result_sizes(mod, x, e_x, u, e_u)" in body, fn


def test_julia_shim_binds_every_entry_point_of_the_header():
    """julia/AlmpcHIP.jl is the binding a maintainer of the reference would add: every symbol include/almpc.h declares has a ccall
    there (the shim cannot be executed here -- no julia -- so at least its coverage of the ABI is checked)."""
    import re
    header = open(os.path.join(ROOT, "include", "almpc.h")).read()
    names = sorted(set(re.findall(r"\b(almpc_[a-z0-9_]+)\s*\(", header)))
    shim = open(os.path.join(ROOT, "julia", "AlmpcHIP.jl")).read()
    bound = set(re.findall(r"\(:(almpc_[a-z0-9_]+), libalmpc\)", shim))
    missing = [n for n in names if n not in bound]
    assert not missing, f"not bound in julia/AlmpcHIP.jl: {missing}"
    assert len(names) >= 45
