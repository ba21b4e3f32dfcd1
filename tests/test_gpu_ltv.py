"""GPU parity tests for time-varying models (almpc_design_ltv): the QP of one SQP / multiple-shooting iteration.  The reference
has no such path (BASELINE.json configs[4]; its NLP methods were removed), so parity is on the QP itself: against the numpy
restatement oracle/mpc_oracle.py::ltv_qp + the exact box-QP solver, and against the time-invariant MPC optimum when the stage
models coincide."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
U_TOL = 1e-5


def solve_ltv(capi, A_all, B_all, c_all, xbar, ubar, x_ref, u_ref, Q, R, S, P, umin, umax, opts=None, **kw):
    b, N, n = A_all.shape[0], A_all.shape[1], A_all.shape[2]
    m = B_all.shape[3]
    s = capi.Solver(n, m, N, b)
    s.design_ltv(A_all, B_all, c_all, xbar, ubar, x_ref, u_ref, Q, R, S, P, umin, umax, **kw)
    s.update_initialization(xbar[:, :, 0])
    s.calculate(opts)
    r = s.get_results(want=("u", "e_u", "status", "iters", "polish_iters"))
    return s, r


def test_ltv_design_and_solution_vs_oracle(capi, mo):
    rng = np.random.default_rng(3)
    b, n, m, N = 11, 5, 2, 12
    A_all = np.stack([[0.9 * np.eye(n) + 0.15 * rng.standard_normal((n, n)) for _ in range(N)] for _ in range(b)])
    B_all = rng.standard_normal((b, N, n, m))
    c_all = 0.1 * rng.standard_normal((b, N, n))
    xbar = rng.standard_normal((b, n, N + 1))
    ubar = 0.3 * rng.standard_normal((b, m, N))
    x_ref = 0.2 * rng.standard_normal((n, N + 1)); u_ref = 0.1 * rng.standard_normal((m, N))
    Q, R, S = 10.0 * np.eye(n), 1.0 * np.eye(m), 0.4 * np.eye(m)
    P = np.stack([(20.0 + i) * np.eye(n) + 0.1 * np.ones((n, n)) for i in range(b)])
    umin, umax = -0.6 * np.ones(m), 0.8 * np.ones(m)
    s, r = solve_ltv(capi, A_all, B_all, c_all, xbar, ubar, x_ref, u_ref, Q, R, S, P, umin, umax)
    nact = 0
    for i in range(b):
        H, q, lo, hi = mo.ltv_qp(A_all[i], B_all[i], c_all[i], xbar[i], ubar[i], x_ref, u_ref, Q, R, S, P[i], umin, umax)
        g = s.get_design_instance(i)
        assert np.abs(g["H"] - H).max() <= 1e-11 * np.abs(H).max()
        assert np.abs(s.get_gradient_instance(i) - q).max() <= 1e-11 * max(1.0, np.abs(q).max())
        v = mo.solve_box_qp_exact(H, q, lo, hi)
        assert r["status"][i] == 0
        assert np.abs(r["e_u"][i].T.reshape(-1) - v).max() <= U_TOL
        assert np.abs(r["u"][i] - (ubar[i] + v.reshape(N, m).T)).max() <= U_TOL
        nact += (np.isclose(v, lo) | np.isclose(v, hi)).sum()
    assert nact > 5
    s.close()


def test_time_invariant_stages_reproduce_the_lti_optimum(capi, mo):
    """A_k = A, B_k = B, zero defects (xbar = the linear rollout of an arbitrary ubar): u = ubar + v is the optimum of the
    reference's QP for (A, B), whatever ubar was."""
    p = mo.quadrotor(N=20)
    b, n, m, N = 24, 12, 4, 20
    X0 = mo.quadrotor_x0_batch(b, 2.0, first_instance=900)
    rng = np.random.default_rng(0)
    ubar = np.clip(0.02 * rng.standard_normal((b, m, N)), p.u_min[None, :, None], p.u_max[None, :, None])
    xbar = np.zeros((b, n, N + 1))
    xbar[:, :, 0] = X0
    for k in range(N):
        xbar[:, :, k + 1] = xbar[:, :, k] @ p.A.T + ubar[:, :, k] @ p.B.T
    A_all = np.broadcast_to(p.A, (b, N, n, n)).copy(); B_all = np.broadcast_to(p.B, (b, N, n, m)).copy()
    s, r = solve_ltv(capi, A_all, B_all, None, xbar, ubar, p.x_ref, p.u_ref, p.Q, p.R, p.S, p.P, p.u_min, p.u_max,
                     opts=capi.default_opts(rho=30.0, max_iter=8, check_every=8), rho=30.0, rho_profile="stiffness")
    s.close()
    assert np.all(r["status"] == 0)
    for i in range(0, b, 3):
        e = mo.solve_mpc_exact(p, X0[i])
        assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL


def test_sqp_iterations_on_the_fnn_model(capi, mo):
    """BASELINE configs[4] in miniature: SQP (multiple shooting, full steps) around the QP engine for the synthetic Fnn model,
    N = 50, batch 256 -> 32 here: Jacobians of every stage by k_fnn_jacobian, the QP by almpc_design_ltv + the per-instance step.
    Every inner QP is checked against the exact solver; the outer iteration drives step and defects to zero."""
    f = mo.synthetic_fnn(act="tanh")    # smooth activation: the reference's NLP branch registers any NNlib activation (.../fnn/...:120-122)
    b, n, m, N = 32, 4, 2, 50
    x_ref = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N + 1)); u_ref = np.tile(np.array([0.1, -0.2])[:, None], (1, N))
    X0 = x_ref[:, 0][None, :] + 0.6 * mo.splitmix_normal(0x5EED0005, 0, b, n)
    Q, R, P = 100.0 * np.eye(n), 0.1 * np.eye(m), 150.0 * np.eye(n)
    umin, umax = -np.ones(m), np.ones(m)
    xbar = np.repeat(X0[:, :, None], N + 1, axis=2)       # initial guess: stay at x0 with the reference input
    ubar = np.repeat(u_ref[None], b, axis=0).copy()
    s = capi.Solver(n, m, N, b)
    hist = []
    for it in range(16):
        pts_x = xbar[:, :, :N].transpose(0, 2, 1).reshape(b * N, n); pts_u = ubar.transpose(0, 2, 1).reshape(b * N, m)
        A, B, fx = capi.fnn_linearize(f.W_in, f.W_h, f.b_h, f.W_out, pts_x, pts_u, act=f.act, want_f=True)
        A_all, B_all = A.reshape(b, N, n, n), B.reshape(b, N, n, m)
        c_all = fx.reshape(b, N, n) - xbar[:, :, 1:].transpose(0, 2, 1)
        s.design_ltv(A_all, B_all, c_all, xbar, ubar, x_ref, u_ref, Q, R, None, P, umin, umax)
        s.update_initialization(X0)
        s.calculate()
        r = s.get_results(want=("e_u", "status"))
        assert np.all(r["status"] == 0)
        V = r["e_u"]
        for i in (0, 7, 19):
            H, q, lo, hi = mo.ltv_qp(A_all[i], B_all[i], c_all[i], xbar[i], ubar[i], x_ref, u_ref, Q, R, 0 * R, P, umin, umax)
            v = mo.solve_box_qp_exact(H, q, lo, hi)
            assert np.abs(V[i].T.reshape(-1) - v).max() <= U_TOL
        # full step: dx from the linearised dynamics (host side of the outer loop)
        dx = np.zeros((b, n))
        xnew = xbar.copy()
        for k in range(N):
            dx = np.einsum("bij,bj->bi", A_all[:, k], dx) + np.einsum("bij,bj->bi", B_all[:, k], V[:, :, k]) + c_all[:, k]
            xnew[:, :, k + 1] = xbar[:, :, k + 1] + dx
        hist.append((float(np.abs(V).max()), float(np.abs(c_all).max())))
        xbar, ubar = xnew, ubar + V
        assert np.all(ubar <= 1 + 1e-9) and np.all(ubar >= -1 - 1e-9)
    s.close()
    # Gauss-Newton Hessian with a non-zero tracking residual: linear convergence of the step, quadratic of the defects
    assert hist[-1][0] <= 1e-4 and hist[-1][1] <= 1e-9, hist
    assert all(hist[k + 1][0] < 0.7 * hist[k][0] for k in range(10, 15)), hist
    assert hist[0][0] > 1e-2


def test_ltv_error_behaviour(capi, mo):
    p = mo.double_integrator()
    b, n, m, N = 3, 2, 1, 10
    s = capi.Solver(n, m, N, b)
    A_all = np.broadcast_to(p.A, (b, N, n, n)).copy(); B_all = np.broadcast_to(p.B, (b, N, n, m)).copy()
    xbar = np.zeros((b, n, N + 1)); ubar = np.zeros((b, m, N))
    with pytest.raises(capi.AlmpcError) as ei:
        s.get_gradient_instance(0)
    assert ei.value.code == -5
    with pytest.raises(Exception):   # P is required
        s.design_ltv(A_all, B_all, None, xbar, ubar, None, None, p.Q, p.R, None, None, p.u_min, p.u_max)
    s.design_ltv(A_all, B_all, None, xbar, ubar, None, None, p.Q, p.R, None, p.P, p.u_min, p.u_max)
    with pytest.raises(capi.AlmpcError) as ei:   # references belong to the design
        s.set_reference(p.x_ref, p.u_ref)
    assert ei.value.code == -1
    s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max)   # back to the shared path: references work again
    s.set_reference(p.x_ref, p.u_ref)
    s.close()
    big = capi.Solver(40, 3, 40, 2)   # 1720 elements of (A, B) per stage: beyond the register build; LDS build: 280 KB of accumulators
    with pytest.raises(capi.AlmpcError) as ei:
        big.design_ltv(np.zeros((2, 40, 40, 40)), np.zeros((2, 40, 40, 3)), None, np.zeros((2, 40, 41)), np.zeros((2, 3, 40)), None, None,
                       np.eye(40), np.eye(3), None, np.eye(40), -np.ones(3), np.ones(3))
    assert ei.value.code == -4
    big.close()


@pytest.mark.parametrize("path", ["registers", "lds"])
@pytest.mark.parametrize("shape", [(12, 4, 32), (3, 1, 37), (6, 3, 20)])
def test_ltv_design_kernels_both_builds(capi, mo, path, shape, monkeypatch):
    """k_design_ltv_reg (accumulators in registers, nz <= 128; (12, 4, 32) is its largest quadrotor-like shape) and the LDS build
    it replaced (still the route for larger nz; forced here through ALMPC_LTV_LDS) against the numpy restatement: H, q and u*."""
    n, m, N = shape
    if path == "lds":
        if ((m * N) ** 2 + 3 * n * m * N + 4 * n * n + n * m + 3 * n + m * N) * 8 > 160 * 1024:
            pytest.skip("LDS build: nz^2 + 3 n nz doubles do not fit 160 KB")
        monkeypatch.setenv("ALMPC_LTV_LDS", "1")
    rng = np.random.default_rng(n * 100 + N)
    b = 5
    A_all = np.stack([[0.85 * np.eye(n) + 0.1 * rng.standard_normal((n, n)) / np.sqrt(n) for _ in range(N)] for _ in range(b)])
    B_all = rng.standard_normal((b, N, n, m))
    c_all = 0.05 * rng.standard_normal((b, N, n))
    xbar = rng.standard_normal((b, n, N + 1)); ubar = 0.2 * rng.standard_normal((b, m, N))
    x_ref = 0.2 * rng.standard_normal((n, N + 1)); u_ref = 0.1 * rng.standard_normal((m, N))
    Q, R, S, P = 10.0 * np.eye(n), 1.0 * np.eye(m), 0.3 * np.eye(m), 25.0 * np.eye(n) + 0.2 * np.ones((n, n))
    umin, umax = -0.5 * np.ones(m), 0.6 * np.ones(m)
    s, r = solve_ltv(capi, A_all, B_all, c_all, xbar, ubar, x_ref, u_ref, Q, R, S, P, umin, umax)
    assert np.all(r["status"] == 0)
    for i in range(b):
        H, q, lo, hi = mo.ltv_qp(A_all[i], B_all[i], c_all[i], xbar[i], ubar[i], x_ref, u_ref, Q, R, S, P, umin, umax)
        Hd = s.get_design_instance(i)["H"]
        assert np.abs(Hd - H).max() <= 1e-11 * np.abs(H).max()
        assert np.abs(s.get_gradient_instance(i) - q).max() <= 1e-11 * max(1.0, np.abs(q).max())
        v = mo.solve_box_qp_exact(H, q, lo, hi)
        assert np.abs(r["e_u"][i].T.reshape(-1) - v).max() <= U_TOL
    s.close()
