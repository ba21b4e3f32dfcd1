"""CPU tests of the stage-wise dual active-set restatement (oracle/stagewise_oracle.py = the algorithm of csrc/almpc_sdual.hip.h) and of
the host Riccati recursion behind the kernel's stage records (hm::stage_records, csrc/almpc_host_math.h, under ASan / UBSan).
Truth: mpc_oracle.solve_mpc_exact / solve_qp_rows_exact -- the condensed exact solvers with their method-independent KKT certificate --
i.e. the reference's QP (..linear.jl:48-100, src/sub/design_mpc.jl:298-468) solved another way."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

XMAX = np.array([3, 3, 3, 1.5, 1.5, 1.5, 0.3, 0.3, 0.3, 1.0, 1.0, 1.0])


@pytest.fixture(scope="module")
def so():
    import stagewise_oracle
    return stagewise_oracle


def test_a_sweep_is_a_column_of_the_condensed_inverse_hessian(mo, so):
    """Ghat[:, j] for an input row j = column j of H^-1 of the condensed QP; for a state row = Gamma H^-1 Gamma_row'."""
    d = mo.double_integrator(10)
    for s_w in (0.0, 3.0):
        p = mo.make_problem(d.A, d.B, 10, d.u_min, d.u_max, s=s_w)
        q = so.stage_qp_from_problem(p, np.array([5.0, 0.0]))
        g = so.stage_gains(q)
        _, Gam, H, _ = mo.condense(p)
        Hi = np.linalg.inv(H)
        for k in (0, 3, 9):
            sx = np.zeros((p.N + 1, q.nt)); su = np.zeros((p.N, p.m)); su[k, 0] = -0.5
            v, x = so.stage_sweep(q, g, sx, su, np.zeros(q.nt), False)
            assert np.abs(v.reshape(-1) - Hi[:, k]).max() <= 1e-13 * np.abs(Hi).max()
            assert np.abs(x[1:, :p.n].reshape(-1) - Gam @ Hi[:, k]).max() <= 1e-12 * np.abs(Gam @ Hi).max()
        for k, i in ((4, 1), (10, 0)):
            sx = np.zeros((p.N + 1, q.nt)); su = np.zeros((p.N, p.m)); sx[k, i] = -0.5
            v, x = so.stage_sweep(q, g, sx, su, np.zeros(q.nt), False)
            row = Gam[(k - 1) * p.n + i]
            assert np.abs(v.reshape(-1) - Hi @ row).max() <= 1e-12 * np.abs(Hi @ row).max()


@pytest.mark.parametrize("case", ["di_all", "quad_box", "quad_eq", "quad_S", "quad_S_box_tvref"])
def test_stagewise_dual_equals_the_exact_solver(mo, so, case):
    if case == "di_all":
        d = mo.double_integrator(10)
        p = mo.make_problem(d.A, d.B, 6, d.u_min, d.u_max, s=2.0, x_min=[-6.0, -1.5], x_max=[6.0, 1.5], terminal="equality")
        rng = np.random.default_rng(3)
        X0 = np.stack([rng.uniform(-5.9, 5.9, 40), rng.uniform(-1.4, 1.4, 40)], axis=1)
    else:
        q = mo.quadrotor(20)
        kw = {}
        if "box" in case:
            kw.update(x_min=-XMAX, x_max=XMAX)
        if case == "quad_eq":
            kw.update(terminal="equality")
        if "_S" in case:
            kw.update(s=5.0)
        if "tvref" in case:
            kw.update(u_ref=0.01 * np.sin(np.arange(20))[None, :] * np.ones((4, 1)))
        p = mo.make_problem(q.A, q.B, 20, q.u_min, q.u_max, **kw)
        X0 = mo.quadrotor_x0_batch(8, 1.0 if case == "quad_eq" else 3.0)
        if p.x_min is not None:
            X0 = np.clip(X0, -0.99 * XMAX, 0.99 * XMAX)
    n_ok = n_inf = 0
    for x0 in X0:
        r = so.solve_mpc_stagewise(p, x0)
        try:
            e = mo.solve_mpc_exact(p, x0)
        except ValueError:
            assert r["status"] == 3
            n_inf += 1
            continue
        assert r["status"] == 0 and np.abs(r["u"] - e["u"]).max() <= 1e-8
        assert np.abs(r["e_solver"] - e["e_x"]).max() <= 1e-7     # the solver's own trajectory = the rollout of its inputs
        n_ok += 1
    assert n_ok >= 3
    if case == "di_all":
        assert n_inf >= 1


def test_a_guess_changes_the_iteration_count_not_the_optimum(mo, so):
    p30, p50 = mo.quadrotor(30), mo.quadrotor(50)
    for x0 in mo.quadrotor_x0_batch(4, 3.0):
        cold = so.solve_mpc_stagewise(p50, x0)
        guess = np.hstack([mo.solve_mpc_exact(p30, x0)["u"], np.zeros((4, 20))])
        warm = so.solve_mpc_stagewise(p50, x0, u_guess=guess)
        assert cold["status"] == 0 and warm["status"] == 0
        assert np.abs(cold["u"] - warm["u"]).max() <= 1e-9 and warm["iters"] <= cold["iters"]


def test_stagewise_dual_solves_the_qp_of_an_sqp_iteration(mo, so):
    """Time-varying stage models, defects, state errors, input gradient, input-rate weight and a state box: the QP of an SQP iteration
    (mpc_oracle.ltv_qp + ltv_state_rows) in its stage-wise form."""
    f = mo.synthetic_fnn(act="tanh")
    n, m, N = 4, 2, 15
    x_ref = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N + 1)); u_ref = np.tile(np.array([0.1, -0.2])[:, None], (1, N))
    Q, R, P, S = 100.0 * np.eye(n), 0.1 * np.eye(m), 150.0 * np.eye(n), 0.4 * np.eye(m)
    umin, umax = -np.ones(m), np.ones(m)
    x0 = x_ref[:, 0] + 0.6 * mo.splitmix_normal(0x5EED0005, 3, 1, n)[0]
    U = np.clip(u_ref + 0.3 * np.sin(np.arange(N))[None, :], umin[:, None], umax[:, None])
    X = mo.fnn_rollout(f, x0, U)
    X[:, 1:] += 0.01 * np.cos(np.arange(N))[None, :]          # non-zero defects
    A, B, c = [], [], []
    for k in range(N):
        Ak, Bk = f.jacobian(X[:, k], U[:, k])
        A.append(Ak); B.append(Bk); c.append(f.forward(X[:, k], U[:, k]) - X[:, k + 1])
    for box in (None, 1.2):
        xmin = None if box is None else -box * np.ones(n)
        xmax = None if box is None else box * np.ones(n)
        q = so.stage_qp_from_ltv(A, B, c, X, U, x_ref, u_ref, Q, R, S, P, umin, umax, x_min=xmin, x_max=xmax)
        r = so.solve_stage_dual(q)
        H, g, lo, hi, Gam, gg = mo.ltv_qp(A, B, c, X, U, x_ref, u_ref, Q, R, S, P, umin, umax, return_prediction=True)
        if box is None:
            v = mo.solve_box_qp_exact(H, g, lo, hi)
        else:
            C, a0, lo_c, hi_c, eq_c = mo.ltv_state_rows(Gam, gg, X, x_ref, xmin, xmax, "none")
            v, _ = mo.solve_qp_rows_exact(H, g, lo, hi, C, a0, lo_c, hi_c, eq_c)
        assert r["status"] == 0 and np.abs(r["v"].reshape(-1) - v).max() <= 1e-8


@pytest.mark.timeout(600)
@pytest.mark.parametrize("useS", [0, 1])
def test_host_stage_records_under_asan_ubsan(tmp_path, mo, so, useS):
    """hm::stage_records = stagewise_oracle.stage_gains, record layout of csrc/almpc_sdual.hip.h; the DARE terminal weight without S
    gives stage-invariant records."""
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
    exe = str(tmp_path / "stage_records_san")
    subprocess.check_call(["g++", "-std=c++17", "-Wall"] + san + ["-o", exe, os.path.join(ROOT, "tests", "sanitize", "stage_records_driver.cpp")])
    q = mo.quadrotor(12)
    p = mo.make_problem(q.A, q.B, 12, q.u_min, q.u_max, s=5.0 if useS else 0.0)
    n, m, N = p.n, p.m, p.N
    nt = n + m if useS else n
    NT, MC = (16, 4) if useS else (12, 4)
    with open(tmp_path / "in.txt", "w") as fo:
        fo.write(f"{n} {m} {N} {useS} {NT} {MC}\n")
        for M in (p.A, p.B, p.Q, p.R, p.S, p.P):
            fo.write(" ".join(repr(float(v)) for v in np.asarray(M).flatten(order="F")) + "\n")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, str(tmp_path / "in.txt")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "ERROR" not in r.stderr and "runtime error" not in r.stderr
    vals = r.stdout.split()
    assert vals[0] == "1" and int(vals[1]) == (0 if useS else 1)
    RL, LRP = 2 * NT + 2 * MC, NT + MC
    rec = np.array([float(v) for v in vals[2:]]).reshape(N, LRP, RL)
    g = so.stage_gains(so.stage_qp_from_problem(p, np.zeros(n)))
    qq = so.stage_qp_from_problem(p, np.zeros(n))
    for k in (0, 1, N // 2, N - 1):
        K, Li, Acl, Lb, Bt = g["K"][k], g["Li"][k], g["Acl"][k], g["Lb"][k], qq.Bt(k)
        sc = max(np.abs(Acl).max(), np.abs(K).max(), np.abs(Lb).max())
        assert np.abs(rec[k, :nt, :nt] - Acl.T).max() <= 1e-9 * sc
        assert np.abs(rec[k, :nt, NT:NT + m] + K.T).max() <= 1e-9 * sc
        assert np.abs(rec[k, :nt, NT + MC:NT + MC + nt] - Acl).max() <= 1e-9 * sc
        assert np.abs(rec[k, :nt, 2 * NT + MC:2 * NT + MC + m] + Bt).max() <= 1e-12
        assert np.abs(rec[k, NT:NT + m, :nt] - Lb).max() <= 1e-9 * sc
        assert np.abs(rec[k, NT:NT + m, NT:NT + m] - Li).max() <= 1e-9 * np.abs(Li).max()
        assert np.abs(rec[k, NT:NT + m, NT + MC:NT + MC + nt] + K).max() <= 1e-9 * sc
        assert np.abs(rec[k, NT:NT + m, 2 * NT + MC:2 * NT + MC + m] + np.eye(m)).max() == 0.0
        assert np.all(rec[k, nt:NT] == 0.0) and np.all(rec[k, NT + m:] == 0.0)
