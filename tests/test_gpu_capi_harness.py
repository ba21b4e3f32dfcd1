"""The C ABI from a non-Python caller: tests/capi_harness.c (plain C, gcc, links libalmpc.so, column-major arrays as `ccall` passes
them) solves the golden cases of tests/golden/ and checks them itself, makes the almpc_comm_* calls of a one-rank job, and solves the
same cases once more as one model per instance with an (inactive) state box: almpc_set_state_box + almpc_design_batched from C."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def _build(tmp_path, capi):
    exe = str(tmp_path / "capi_harness")
    libdir = os.path.dirname(capi.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "capi_harness.c"),
                           "-L" + libdir, "-lalmpc", "-Wl,-rpath," + libdir, "-lm"])
    return exe


@pytest.mark.parametrize("name", ["double_integrator", "quadrotor"])
def test_c_caller_solves_the_golden_cases(tmp_path, capi, name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        g = json.load(f)
    n, m, N = g["n"], g["m"], g["N"]
    assert g["s"] == 0.0
    A, B = np.array(g["A"]), np.array(g["B"])
    xr = np.array(g["x_ref"], dtype=float).reshape(n, -1) * np.ones((n, N + 1))
    ur = np.array(g["u_ref"], dtype=float).reshape(m, -1) * np.ones((m, N))
    cases = g["cases"]
    cols = lambda M: np.asarray(M, dtype=float).flatten(order="F")   # column-major = the memory of a Julia Matrix{Float64}
    parts = [cols(A), cols(B), cols(g["q"] * np.eye(n)), cols(g["r"] * np.eye(m)), np.array(g["u_min"], float), np.array(g["u_max"], float),
             cols(xr), cols(ur)]
    parts += [np.array(c["x0"], float) for c in cases]
    parts += [cols(np.array(c["u"])) for c in cases]      # (m, N) column-major = [N][m]
    parts += [cols(np.array(c["x"])) for c in cases]      # (n, N+1) column-major = [N+1][n]
    case_file = tmp_path / "case.txt"
    with open(case_file, "w") as f:
        f.write(f"{n} {m} {N} {len(cases)}\n")
        f.write("\n".join(" ".join(repr(float(v)) for v in p) for p in parts) + "\n")
    exe = _build(tmp_path, capi)
    r = subprocess.run([exe, str(case_file)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith(("OK ", "FAIL "))][-1]   # (librccl may print a version banner)
    assert line.startswith("OK ") and f"instances {len(cases)} unsolved 0" in line
    assert "comm_rc 0 ranks 1" in line, line      # the library loaded librccl and ran its collectives on a one-rank communicator
    assert "per-instance+state-box unsolved 0" in line, line
