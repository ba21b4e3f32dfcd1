"""State rows (state box of kw mpc_state_constraint, terminal equality) with a model PER INSTANCE (pytest -m gpu):
almpc_set_state_box / almpc_set_terminal_equality + almpc_design_batched and the re-linearisation pipeline.  The reference adds the
state box in every delegate (…/linear/mpc_modeler_implementation_linear.jl:62-70, …/fnn/mpc_modeler_implementation_fnn.jl:146-153) and
the terminal equality in src/sub/design_mpc.jl:330-331; given (A_i, B_i) instance i's QP is the reference's QP, so the checker is the
exact oracle on that instance's own problem."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

U_TOL, X_TOL = 1e-6, 1e-6


def _check_instances(mo, r, problems, X0, idx):
    n_state_active = n_infeasible = 0
    for i in idx:
        p = problems(i)
        try:
            e = mo.solve_mpc_exact(p, X0[i], return_info=True)
        except ValueError:
            assert r["status"][i] == 3, (i, r["status"][i])
            n_infeasible += 1
            continue
        assert r["status"][i] == 0, (i, r["status"][i])
        assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL, i
        assert np.abs(r["x"][i] - e["x"]).max() <= X_TOL * max(1.0, np.abs(e["x"]).max()), i
        n_state_active += e["info"]["n_active_state"]
        if p.x_min is not None:
            assert np.all(r["x"][i] <= p.x_max[:, None] + 1e-9) and np.all(r["x"][i] >= p.x_min[:, None] - 1e-9)
        if p.terminal == "equality":
            assert np.abs(r["e_x"][i][:, -1]).max() <= 1e-9
    return n_state_active, n_infeasible


@pytest.mark.parametrize("case", ["box", "eq", "box_eq"])
def test_double_integrator_family_with_state_rows(capi, mo, case):
    """Per-instance sampling times of the double integrator (every instance its own (A_i, B_i)), velocity box and / or terminal
    equality: equal to the shared-model state-row path where the models coincide, and to the exact oracle everywhere."""
    batch, N = 48, 10
    Ts = 0.5 + 0.02 * np.arange(batch)
    A = np.stack([np.array([[1.0, t], [0.0, 1.0]]) for t in Ts])
    B = np.stack([np.array([[0.5 * t * t], [t]]) for t in Ts])
    xmin, xmax = (None, None) if case == "eq" else (np.array([-10.0, -0.8]), np.array([10.0, 0.8]))
    terminal = "none" if case == "box" else "equality"
    X0 = mo.splitmix_normal(0x5EED0011, 0, batch, 2) * np.array([2.0, 0.4])[None]
    if xmin is not None:
        X0[:, 1] = np.clip(X0[:, 1], -0.75, 0.75)
        X0[3, 1] = 0.9   # outside the box at stage 1: infeasible, as the reference's x[:,1] = x0 constraint makes it
    s = capi.Solver(2, 1, N, batch)
    s.design_batched(A, B, 100.0 * np.eye(2), 0.1 * np.eye(1), None, None, [-1.0], [1.0], xmin=xmin, xmax=xmax, terminal=terminal)
    s.update_initialization(X0)
    s.debug_poison_lds()
    s.calculate()
    r = s.get_results()
    s.close()
    probs = lambda i: mo.make_problem(A[i], B[i], N, [-1.0], [1.0], x_min=xmin, x_max=xmax, terminal=terminal)
    na, ninf = _check_instances(mo, r, probs, X0, range(batch))
    assert na > 0, "test inputs never activate a state row"
    if xmin is not None:
        assert ninf >= 1


def test_quadrotor_family_tight_box_and_equality(capi, mo):
    """Quadrotor models with per-instance mass (B scaled), tight state box + terminal equality: working sets beyond 32 rows go
    through the 64-row build with per-instance constraint-space matrices."""
    q = mo.quadrotor()
    batch, N = 24, 30
    scale = 1.0 + 0.05 * np.sin(np.arange(batch))
    A = np.repeat(q.A[None], batch, 0)
    B = q.B[None] * scale[:, None, None]
    xmax = np.array([1, 1, 1, .5, .5, .5, .1, .1, .1, .1, .1, .1]) * 3.0
    X0 = np.clip(mo.quadrotor_x0_batch(batch, 1.0, first_instance=900), -0.99 * xmax, 0.99 * xmax)
    s = capi.Solver(12, 4, N, batch)
    s.design_batched(A, B, q.Q, q.R, None, q.P, q.u_min, q.u_max, xmin=-xmax, xmax=xmax, terminal="equality")
    s.update_initialization(X0)
    s.calculate()
    r = s.get_results()
    s.close()
    probs = lambda i: mo.make_problem(A[i], B[i], N, q.u_min, q.u_max, x_min=-xmax, x_max=xmax, terminal="equality", P=q.P)
    na, _ = _check_instances(mo, r, probs, X0, range(batch))
    assert na > 12 * 4


def test_relin_pipeline_with_state_box(capi, mo):
    """Re-linearisation pipeline (BASELINE configs[3] shape) with the state box of kw mpc_state_constraint: every instance's QP is the
    reference's LP-branch QP for its own linearisation, including the box rows of …/fnn/mpc_modeler_implementation_fnn.jl:52-58."""
    f = mo.synthetic_fnn()
    batch, N, n, m = 128, 20, 4, 2
    x_ref = np.array([0.2, -0.1, 0.05, 0.0])[:, None] * np.ones((n, N + 1))
    u_ref = np.array([0.1, -0.2])[:, None] * np.ones((m, N))
    Q, R = 100.0 * np.eye(n), 0.1 * np.eye(m)
    Al, Bl = capi.fnn_linearize(f.W_in, f.W_h, f.b_h, f.W_out, x_ref[:, -1][None], u_ref[:, -1][None], act=f.act)
    P = capi.dare(Al[0], Bl[0], Q, R)
    xmax = np.array([1.2, 1.2, 1.2, 1.2])
    X0 = np.clip(x_ref[:, 0][None, :] + mo.splitmix_normal(0x5EED0004, 31, batch, n) * 0.8, -0.98 * xmax, 0.98 * xmax)
    s = capi.Solver(n, m, N, batch, timing=True)
    s.relin_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, None, P, [-1, -1], [1, 1], act=f.act, xmin=-xmax, xmax=xmax)
    s.update_initialization(0.7 * X0)
    s.relin_fnn_step()
    s.update_initialization(X0)        # a second step from other states: new Jacobians, new designs, new constraint-space matrices
    s.relin_fnn_step()
    r = s.get_results()
    s.close()

    def probs(i):
        Ai, Bi = f.jacobian(X0[i], u_ref[:, 0])
        return mo.make_problem(Ai, Bi, N, [-1, -1], [1, 1], x_ref=x_ref, u_ref=u_ref, x_min=-xmax, x_max=xmax, P=P)
    # (unstable linearisations: compare where the prediction stays in a meaningful range)
    idx = [i for i in range(0, batch, 3) if np.abs(np.linalg.eigvals(f.jacobian(X0[i], u_ref[:, 0])[0])).max() < 1.3]
    na, ninf = _check_instances(mo, r, probs, X0, idx)
    assert len(idx) > 20 and na > 0
    # a warm step (guess = the previous inputs shifted, no ADMM phase) from the same states lands on the same optima
    s2 = capi.Solver(n, m, N, batch)
    s2.relin_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, None, P, [-1, -1], [1, 1], act=f.act, xmin=-xmax, xmax=xmax)
    s2.update_initialization(0.9 * X0)
    s2.relin_fnn_step()
    s2.update_initialization(X0)
    s2.relin_fnn_step(capi.default_opts(warm_start=1))
    r2 = s2.get_results()
    s2.close()
    assert np.all(r2["iters"] == 0)
    same = (r["status"] == 0) & (r2["status"] == 0)
    assert np.array_equal(r["status"] == 3, r2["status"] == 3) and same.sum() >= 0.9 * batch
    assert np.abs(r["u"][same] - r2["u"][same]).max() <= U_TOL


def _ltv_case(mo, b, N, seed):
    """A time-varying QP per instance: double integrators whose sampling time changes from stage to stage (A_k, B_k), a linearisation
    trajectory xbar that is NOT a rollout (non-zero defects c_k, as in multiple shooting) and a guess ubar."""
    n, m = 2, 1
    x_ref = np.zeros((n, N + 1)); u_ref = np.zeros((m, N))
    X0 = mo.splitmix_normal(seed, 0, b, n) * np.array([1.5, 0.3])[None]
    Ug = 0.2 * mo.splitmix_normal(seed + 1, 0, b, m * N).reshape(b, m, N)
    Ts = 0.5 + 0.1 * np.abs(mo.splitmix_normal(seed + 3, 0, b, N))
    A = np.zeros((b, N, n, n)); B = np.zeros((b, N, n, m)); c = np.zeros((b, N, n)); Xb = np.zeros((b, n, N + 1))
    for i in range(b):
        X = np.zeros((n, N + 1)); X[:, 0] = X0[i]
        for k in range(N):
            t = Ts[i, k]
            A[i, k] = np.array([[1.0, t], [0.0, 1.0]]); B[i, k] = np.array([[0.5 * t * t], [t]])
            X[:, k + 1] = A[i, k] @ X[:, k] + B[i, k] @ Ug[i][:, k]
        X[:, 1:] *= 0.9                                   # (off the rollout: defects)
        Xb[i] = X
        for k in range(N):
            c[i, k] = A[i, k] @ X[:, k] + B[i, k] @ Ug[i][:, k] - X[:, k + 1]
    return x_ref, u_ref, X0, Ug, A, B, c, Xb


@pytest.mark.parametrize("case", ["box", "box_eq"])
def test_ltv_design_with_state_rows(capi, mo, case):
    """almpc_design_ltv with the state box (and the terminal equality): rows of dX = Gam v + g with bounds relative to the
    linearisation trajectory; checker: the exact dual active set on the same QP (oracle ltv_qp + ltv_state_rows)."""
    b, N, n, m = 32, 12, 2, 1
    x_ref, u_ref, X0, Ug, A, B, c, Xb = _ltv_case(mo, b, N, 0x5EED0021)
    Q, R, S, P = 100.0 * np.eye(n), 0.1 * np.eye(m), np.zeros((m, m)), 150.0 * np.eye(n)
    umin, umax = -np.ones(m), np.ones(m)
    xmax = np.array([10.0, 0.6])
    terminal = "equality" if case == "box_eq" else "none"
    s = capi.Solver(n, m, N, b)
    s.design_ltv(A, B, c, Xb, Ug, x_ref, u_ref, Q, R, S, P, umin, umax, xmin=-xmax, xmax=xmax, terminal=terminal)
    s.calculate()
    r = s.get_results(want=("status", "u", "e_u"))
    s.close()
    nact = ninf = 0
    for i in range(b):
        Al, Bl, cl = list(A[i]), list(B[i]), list(c[i])
        H, q, lo, hi, Gam, g = mo.ltv_qp(Al, Bl, cl, Xb[i], Ug[i], x_ref, u_ref, Q, R, S, P, umin, umax, return_prediction=True)
        C, a0, lo_c, hi_c, eq_c = mo.ltv_state_rows(Gam, g, Xb[i], x_ref, -xmax, xmax, terminal)
        try:
            if np.any(np.abs(Xb[i][:, 0]) > xmax):
                raise ValueError("x0 outside")
            v, W = mo.solve_qp_rows_exact(H, q, lo, hi, C, a0, lo_c, hi_c, eq_c)
        except ValueError:
            assert r["status"][i] == 3, (i, r["status"][i])
            ninf += 1
            continue
        assert r["status"][i] == 0, (i, r["status"][i])
        assert np.abs(r["e_u"][i].T.reshape(-1) - v).max() <= U_TOL, i
        dX = Gam @ v + g
        assert np.all(np.abs(Xb[i][:, 1:].T.reshape(-1) + dX) <= np.tile(xmax, N) + 1e-8)
        if terminal == "equality":
            assert np.abs((Xb[i][:, 1:].T.reshape(-1) + dX)[-n:] - x_ref[:, N]).max() <= 1e-8
        nact += sum(1 for j in W if j >= m * N and not eq_c[j - m * N])
    assert nact > 0 and ninf < b // 2, (nact, ninf)


def test_mirror_state_constraint_in_step_and_nlp_modes(pkg, capi, mo):
    """Host mirror: kw mpc_state_constraint with a black-box model -- mpc_linearization='step' (per-instance LP-branch QPs) and
    mpc_programming_type='non_linear' (SQP): the rows the reference adds at .../fnn/mpc_modeler_implementation_fnn.jl:52-58,146-153."""
    f = mo.synthetic_fnn(act="tanh")
    n, m, N, batch = 4, 2, 20, 12
    xlo, xhi = np.array([-0.12, -0.58, -0.25, -0.35]), np.array([0.25, 0.09, 0.07, 0.22])
    sys_ = pkg.ConstrainedBlackBoxControlDiscreteSystem(pkg.Fnn(f.W_in, f.W_h, f.b_h, f.W_out, f.act), n, m,
                                                        pkg.Hyperrectangle(xlo, xhi), pkg.Hyperrectangle([-1, -1], [1, 1]))
    x_ref, u_ref = [0.2, -0.1, 0.05, 0.0], [0.1, -0.2]
    X0 = np.asarray(x_ref)[None, :] + 0.5 * mo.splitmix_normal(0x5EED0005, 40, batch, n)
    X0 = np.clip(X0, xlo + 0.02 * (xhi - xlo), xhi - 0.02 * (xhi - xlo))
    keep = [i for i in range(batch) if i != 9]      # (instance 9: infeasible first QP, see tests/test_gpu_sqp.py)
    X0 = X0[keep + [0]]                             # still 12 instances
    # step mode (the linearisation at x0 is a cruder model: four of these QPs are infeasible in the tight box, so it is widened by 0.05)
    sys_w = pkg.ConstrainedBlackBoxControlDiscreteSystem(pkg.Fnn(f.W_in, f.W_h, f.b_h, f.W_out, f.act), n, m,
                                                         pkg.Hyperrectangle(xlo - 0.05, xhi + 0.05), pkg.Hyperrectangle([-1, -1], [1, 1]))
    C = pkg.proceed_controller(sys_w, "model_predictive_control", N, 1, x_ref, u_ref, mpc_batch=batch, mpc_linearization="step",
                               mpc_state_constraint=True)
    P = C.tuning.terminal_ingredient.P
    res = pkg._model_predictive_control_computation(C, X0)
    st = C.tuning.modeler.last_status
    assert np.all(st == 0)
    xr = np.asarray(x_ref)[:, None] * np.ones((n, N + 1)); ur = np.asarray(u_ref)[:, None] * np.ones((m, N))
    na = 0
    for i in range(batch):
        Ai, Bi = f.jacobian(X0[i], np.asarray(u_ref))
        p = mo.make_problem(Ai, Bi, N, [-1, -1], [1, 1], x_ref=xr, u_ref=ur, x_min=xlo - 0.05, x_max=xhi + 0.05, P=P)
        e = mo.solve_mpc_exact(p, X0[i], return_info=True)
        assert np.abs(res.u[i] - e["u"]).max() <= U_TOL
        na += e["info"]["n_active_state"]
    assert na > 0
    C.tuning.modeler.solver.close()
    # non-linear programming
    C2 = pkg.proceed_controller(sys_, "model_predictive_control", N, 1, x_ref, u_ref, mpc_batch=batch, mpc_programming_type="non_linear",
                                mpc_state_constraint=True, mpc_sqp_iterations=25, mpc_sqp_step_rule="fixed")
    res2 = pkg._model_predictive_control_computation(C2, X0)
    P2 = C2.tuning.terminal_ingredient.P
    assert np.all(C2.tuning.modeler.last_status == 0)
    assert np.all(res2.x <= xhi[None, :, None] + 1e-6) and np.all(res2.x >= xlo[None, :, None] - 1e-6)
    i = 2
    X, U, hist, na2 = mo.sqp_fnn(f, X0[i], xr, ur, 100.0 * np.eye(n), 0.1 * np.eye(m), np.zeros((m, m)), P2, -np.ones(m), np.ones(m), 25,
                                 x_min=xlo, x_max=xhi, return_active=True)
    assert na2 > 0 and np.abs(res2.u[i] - U).max() <= 1e-5
    C2.tuning.modeler.solver.close()
