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
