"""GPU tests of the device-resident SQP loop for black-box Fnn models (almpc_sqp_fnn_*; BASELINE.json configs[4]).  The problem is
the reference's NonLinearProgramming branch for Fnn (.../fnn/mpc_modeler_implementation_fnn.jl:73-189), which it gives to Ipopt;
neither is runnable here, so the checks are (i) the numpy restatement of the same loop with exact QP solves, iteration by
iteration, and (ii) a method-independent first-order certificate of the NLP itself (adjoint gradient, projected)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
U_TOL = 1e-5


def _setup(capi, mo, b, N, act="tanh", S=None, amp=0.6):
    f = mo.synthetic_fnn(act=act)
    n, m = 4, 2
    x_ref = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N + 1)); u_ref = np.tile(np.array([0.1, -0.2])[:, None], (1, N))
    X0 = x_ref[:, 0][None, :] + amp * mo.splitmix_normal(0x5EED0005, 0, b, n)
    Q, R, P = 100.0 * np.eye(n), 0.1 * np.eye(m), 150.0 * np.eye(n)
    S = np.zeros((m, m)) if S is None else S
    umin, umax = -np.ones(m), np.ones(m)
    s = capi.Solver(n, m, N, b)
    s.sqp_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, S, P, umin, umax, act=act)
    return f, s, dict(x_ref=x_ref, u_ref=u_ref, Q=Q, R=R, S=S, P=P, u_min=umin, u_max=umax), X0


def test_config5_sqp_vs_restatement_and_nlp_certificate(capi, mo):
    """N = 50, Fnn 4-2-16x2 (tanh), 32 instances: 30 full-step iterations on the device."""
    b, N, iters = 32, 50, 30
    f, s, kw, X0 = _setup(capi, mo, b, N)
    s.sqp_fnn_start(X0)
    st, de = s.sqp_fnn_iterate(iters)
    r = s.get_results()
    s.close()
    assert np.all(r["status"] == 0)
    assert st[0] > 1e-2 and st[-1] <= 1e-7 and de[-1] <= 1e-12, (st, de)
    for i in (0, 7, 19, 31):
        X, U, hist = mo.sqp_fnn(f, X0[i], kw["x_ref"], kw["u_ref"], kw["Q"], kw["R"], kw["S"], kw["P"], kw["u_min"], kw["u_max"], iters)
        assert np.abs(r["u"][i] - U).max() <= U_TOL
        assert np.abs(r["x"][i] - X).max() <= 1e-5
    for i in range(b):   # every instance: KKT point of the NLP, trajectory consistent with the network
        U = r["u"][i]
        assert np.abs(r["x"][i] - mo.fnn_rollout(f, X0[i], U)).max() <= 1e-9
        assert mo.nlp_kkt_residual(f, X0[i], U, kw["x_ref"], kw["u_ref"], kw["Q"], kw["R"], kw["S"], kw["P"], kw["u_min"], kw["u_max"]) <= 1e-5
        assert np.all(U <= 1.0) and np.all(U >= -1.0)
    assert np.abs(r["e_u"] - (r["u"] - kw["u_ref"][None])).max() <= 1e-15
    assert np.abs(r["e_x"] - (r["x"] - kw["x_ref"][None])).max() <= 1e-15


def test_sqp_histories_match_the_restatement_per_iteration(capi, mo):
    """Batch of one... of three: the device's per-iteration step / defect norms are maxima over the batch of the restatement's."""
    b, N, iters = 3, 20, 6
    f, s, kw, X0 = _setup(capi, mo, b, N, S=0.2 * np.eye(2))
    ug = 0.4 * mo.splitmix_normal(0x5EED0008, 0, b, 2 * N).reshape(b, 2, N)
    s.sqp_fnn_start(X0, ug)
    st, de = s.sqp_fnn_iterate(iters, step_scale=0.75)
    r = s.get_results(want=("u", "x"))
    s.close()
    H = []
    for i in range(b):
        X, U, hist = mo.sqp_fnn(f, X0[i], kw["x_ref"], kw["u_ref"], kw["Q"], kw["R"], kw["S"], kw["P"], kw["u_min"], kw["u_max"], iters,
                                u_guess=ug[i], step_scale=0.75)
        H.append(hist)
        assert np.abs(r["u"][i] - U).max() <= U_TOL and np.abs(r["x"][i] - X).max() <= 1e-5
    H = np.array(H)   # (b, iters, 2)
    np.testing.assert_allclose(st, H[:, :, 0].max(axis=0), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(de, H[:, :, 1].max(axis=0), rtol=1e-4, atol=1e-9)


def test_sqp_continues_across_calls_and_restarts(capi, mo):
    b, N = 8, 20
    f, s, kw, X0 = _setup(capi, mo, b, N)
    s.sqp_fnn_start(X0)
    s.sqp_fnn_iterate(5)
    s.sqp_fnn_iterate(7)
    u12 = s.get_results(want=("u",))["u"].copy()
    s.sqp_fnn_start(X0)
    s.sqp_fnn_iterate(12)
    u12b = s.get_results(want=("u",))["u"]
    assert np.abs(u12 - u12b).max() == 0.0     # same kernels, same inputs: bit-identical
    s.close()


def test_sqp_error_behaviour(capi, mo):
    f = mo.synthetic_fnn(act="tanh")
    s = capi.Solver(4, 2, 10, 4)
    with pytest.raises(capi.AlmpcError) as ei:
        s.sqp_fnn_start(np.zeros((4, 4)))
    assert ei.value.code == -5
    with pytest.raises(capi.AlmpcError) as ei:
        s.sqp_fnn_iterate(1)
    assert ei.value.code == -5
    s.close()
    f2, s, kw, X0 = _setup(capi, mo, 4, 10)
    with pytest.raises(capi.AlmpcError):
        s.sqp_fnn_iterate(1)                    # not started
    s.sqp_fnn_start(X0)
    with pytest.raises(capi.AlmpcError):
        s.sqp_fnn_iterate(0)
    with pytest.raises(capi.AlmpcError):
        s.sqp_fnn_iterate(1, step_scale=1.5)
    with pytest.raises(capi.AlmpcError):
        s.set_reference(kw["x_ref"], kw["u_ref"])   # references are part of the SQP set-up
    # a later shared design takes the handle back; the SQP state is gone
    p = mo.make_problem(np.eye(4) * 0.9, np.ones((4, 2)) * 0.1, 10, -np.ones(2), np.ones(2))
    s.design_shared(p.A, p.B, p.Q, p.R, None, None, p.u_min, p.u_max)
    with pytest.raises(capi.AlmpcError) as ei:
        s.sqp_fnn_start(X0)
    assert ei.value.code == -5
    s.close()


def test_sqp_indefinite_instance_is_contained(capi, mo):
    """One instance whose condensed Hessian is not positive definite (here through an indefinite terminal weight; in practice an
    open-loop unstable linearisation over a long horizon, cond(H) beyond 1e16) keeps its start and is reported; the other
    instances are bit-identical to a clean run."""
    b, N = 8, 20
    f, s, kw, X0 = _setup(capi, mo, b, N)
    s.sqp_fnn_start(X0)
    s.sqp_fnn_iterate(4)
    clean = s.get_results(want=("u", "x"))
    assert s.sqp_fnn_skipped().sum() == 0
    P = np.repeat(kw["P"][None], b, axis=0).copy()
    P[3] = -1e4 * np.eye(4)
    s.sqp_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, kw["x_ref"], kw["u_ref"], kw["Q"], kw["R"], kw["S"], P, kw["u_min"], kw["u_max"], act="tanh")
    s.sqp_fnn_start(X0)
    start_u = np.clip(kw["u_ref"], -1, 1)
    with pytest.raises(capi.AlmpcError) as ei:
        s.sqp_fnn_iterate(4)
    assert ei.value.code == -6 and "instance 3" in str(ei.value)
    sk = s.sqp_fnn_skipped()
    r = s.get_results(want=("u", "x"))
    st, de = s.sqp_last
    s.close()
    assert list(sk) == [0, 0, 0, 1, 0, 0, 0, 0]
    keep = [i for i in range(b) if i != 3]
    assert np.abs(r["u"][keep] - clean["u"][keep]).max() == 0.0 and np.abs(r["x"][keep] - clean["x"][keep]).max() == 0.0
    assert np.abs(r["u"][3] - start_u).max() == 0.0
    assert np.abs(r["x"][3] - mo.fnn_rollout(f, X0[3], start_u)).max() <= 1e-12
    assert np.all(np.isfinite(st)) and np.all(np.isfinite(de)) and st[0] > 0


def test_mirror_non_linear_programming_type(pkg, mo):
    """proceed_controller(...; mpc_programming_type = "non_linear") for an Fnn system: the reference builds the NLP of
    .../fnn/...:73-189 for Ipopt; the mirror runs the device SQP loop on the same problem (P = DARE at the last reference)."""
    f = mo.synthetic_fnn(act="tanh")
    sys_ = pkg.ConstrainedBlackBoxControlDiscreteSystem(pkg.Fnn(f.W_in, f.W_h, f.b_h, f.W_out, f.act), 4, 2,
                                                        pkg.Hyperrectangle([-10] * 4, [10] * 4), pkg.Hyperrectangle([-1, -1], [1, 1]))
    x_ref, u_ref = [0.2, -0.1, 0.05, 0.0], [0.1, -0.2]
    N, batch = 20, 16
    C = pkg.proceed_controller(sys_, "model_predictive_control", N, 1, x_ref, u_ref, mpc_batch=batch, mpc_programming_type="non_linear",
                               mpc_sqp_iterations=25)
    Al, Bl = f.jacobian(np.array(x_ref), np.array(u_ref))
    P = mo.dare(Al, Bl, 100 * np.eye(4), 0.1 * np.eye(2))
    assert np.abs(C.tuning.terminal_ingredient.P - P).max() <= 1e-9 * np.abs(P).max()
    X0 = np.asarray(x_ref)[None, :] + 0.6 * mo.splitmix_normal(0x5EED0009, 0, batch, 4)
    res = pkg._model_predictive_control_computation(C, X0)
    assert res.u.shape == (batch, 2, N) and res.x.shape == (batch, 4, N + 1)
    xr, ur = np.tile(np.array(x_ref)[:, None], (1, N + 1)), np.tile(np.array(u_ref)[:, None], (1, N))
    for i in range(batch):
        assert mo.nlp_kkt_residual(f, X0[i], res.u[i], xr, ur, 100 * np.eye(4), 0.1 * np.eye(2), np.zeros((2, 2)), P, -np.ones(2), np.ones(2)) <= 1e-5
        assert np.abs(res.x[i] - mo.fnn_rollout(f, X0[i], res.u[i])).max() <= 1e-8
    # closed loop: plant = the network; the shifted warm start needs fewer iterations to the same tolerance
    C.tuning.modeler.sqp["iterations"] = 8
    X1 = np.stack([f.forward(X0[i], res.u[i][:, 0]) for i in range(batch)])
    res = pkg._model_predictive_control_computation(C, X1)
    st, de = C.tuning.modeler.last_sqp_history
    assert st[-1] <= 1e-3 and de[-1] <= 1e-8
    with pytest.raises(NotImplementedError):   # a quadratic constraint, not a QP row (src/sub/design_mpc.jl:333-340)
        pkg.proceed_controller(sys_, "model_predictive_control", N, 1, x_ref, u_ref, mpc_programming_type="non_linear",
                               mpc_terminal_ingredient="contractive")
    # the terminal equality is a row of every iteration's QP (tests/test_gpu_state_rows_instances.py has the solves)
    Ce = pkg.proceed_controller(sys_, "model_predictive_control", N, 1, x_ref, u_ref, mpc_programming_type="non_linear",
                                mpc_terminal_ingredient="equality")
    assert Ce.tuning.terminal_ingredient.Xf == "equality"
    Ce.tuning.modeler.solver.close()
    C.tuning.modeler.solver.close()


def test_golden_sqp_vectors_through_the_c_abi(capi, mo):
    """tests/golden/fnn_sqp.json (certified KKT points of the NLP): 60 device iterations from the same start reach them."""
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fnn_sqp.json")) as fh:
        g = json.load(fh)
    n, m, N = g["n"], g["m"], g["N"]
    X0 = np.array([c["x0"] for c in g["cases"]])
    xr = np.tile(np.array(g["x_ref"])[:, None], (1, N + 1)); ur = np.tile(np.array(g["u_ref"])[:, None], (1, N))
    s = capi.Solver(n, m, N, len(X0))
    s.sqp_fnn_setup(np.array(g["W_in"]), [np.array(w) for w in g["W_h"]], [np.array(b) for b in g["b_h"]], np.array(g["W_out"]), xr, ur,
                    g["q"] * np.eye(n), g["r"] * np.eye(m), g["s"] * np.eye(m), np.array(g["P"]), np.array(g["u_min"]), np.array(g["u_max"]),
                    act=g["act"])
    s.sqp_fnn_start(X0)
    st, de = s.sqp_fnn_iterate(60)
    r = s.get_results(want=("u", "x", "status"))
    s.close()
    assert np.all(r["status"] == 0) and st[-1] <= 1e-8
    for i, c in enumerate(g["cases"]):
        assert np.abs(r["u"][i] - np.array(c["u"])).max() <= U_TOL
        assert np.abs(r["x"][i] - np.array(c["x"])).max() <= 1e-5


def test_merit_safeguard_breaks_the_cycle(capi, mo):
    """Instances 42, 50 and 115 of the benchmark set (N = 50) never settle with full steps -- 115 sits in a two-cycle with
    |v|_inf = 0.0666 -- and converge under step rule 1 (l1 merit function, almpc_sqp_fnn_set_step_rule); instances 0, 35 and 69
    converge either way (35 and 69 slowly: the rule must not hold them back).  Device loop against the restatement with the same
    rule, then the NLP certificate."""
    N, iters = 50, 60
    f = mo.synthetic_fnn(act="tanh")
    n, m = 4, 2
    x_ref = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N + 1)); u_ref = np.tile(np.array([0.1, -0.2])[:, None], (1, N))
    X0all = x_ref[:, 0][None, :] + 0.6 * mo.splitmix_normal(0x5EED0005, 0, 256, n)
    pick = [0, 35, 42, 50, 69, 115]
    X0 = X0all[pick]
    Al, Bl = f.jacobian(x_ref[:, -1], u_ref[:, -1])
    Q, R, S = 100.0 * np.eye(n), 0.1 * np.eye(m), np.zeros((m, m))
    P = mo.dare(Al, Bl, Q, R)
    umin, umax = -np.ones(m), np.ones(m)
    s = capi.Solver(n, m, N, len(pick))
    s.sqp_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, S, P, umin, umax, act="tanh")
    s.sqp_fnn_start(X0)
    st_full, _ = s.sqp_fnn_iterate(iters)
    assert st_full[-1] > 1e-2                     # the cycle
    s.sqp_fnn_start(X0)
    st, de = s.sqp_fnn_iterate(iters, step_rule="merit")
    r = s.get_results(want=("u", "x", "status"))
    s.close()
    # no cycle left: the batch maximum keeps falling (instance 42 converges at a linear rate of 0.9: 8e-4 after 60 iterations)
    assert st[-1] <= 2e-3 and de[-1] <= 1e-7 and np.all(st[-10:] < st[-11:-1]), (st[-12:], de[-5:])
    for j, i in enumerate(pick):
        X, U, hist = mo.sqp_fnn(f, X0all[i], x_ref, u_ref, Q, R, S, P, umin, umax, iters, adaptive=True)
        assert np.abs(r["u"][j] - U).max() <= U_TOL and np.abs(r["x"][j] - X).max() <= 1e-5, i
        if i in (0, 35, 69, 115):
            assert mo.nlp_kkt_residual(f, X0all[i], r["u"][j], x_ref, u_ref, Q, R, S, P, umin, umax) <= 1e-3, i


def test_config5_at_the_benchmark_batch_size(capi, mo):
    """BASELINE configs[4] at the size bench.py runs it (256 instances, Fnn 4-2-16x2 tanh, N = 50, merit-function step rule, 40
    iterations): every instance's last QP solved, trajectories consistent with the network, the NLP's first-order certificate on a
    sample, and the device loop against the restatement on two instances."""
    b, N, iters = 256, 50, 40
    f, s, kw, X0 = _setup(capi, mo, b, N)
    s.sqp_fnn_start(X0)
    st, de = s.sqp_fnn_iterate(iters, step_rule="merit")
    r = s.get_results()
    s.close()
    assert np.all(r["status"] == 0)
    assert np.all(np.isfinite(st)) and np.all(np.isfinite(de)) and de[-1] <= 1e-5
    assert np.all(r["u"] <= 1.0) and np.all(r["u"] >= -1.0)
    worst = 0.0
    for i in range(0, b, 16):
        assert np.abs(r["x"][i] - mo.fnn_rollout(f, X0[i], r["u"][i])).max() <= 1e-5
        worst = max(worst, mo.nlp_kkt_residual(f, X0[i], r["u"][i], kw["x_ref"], kw["u_ref"], kw["Q"], kw["R"], kw["S"], kw["P"], kw["u_min"], kw["u_max"]))
    assert worst <= 5e-2        # the slowest instance of this batch converges linearly at rate 0.9: most are at 1e-9
    for i in (3, 200):
        X, U, hist = mo.sqp_fnn(f, X0[i], kw["x_ref"], kw["u_ref"], kw["Q"], kw["R"], kw["S"], kw["P"], kw["u_min"], kw["u_max"], iters, adaptive=True)
        assert np.abs(r["u"][i] - U).max() <= 1e-5


def test_sqp_with_state_box(capi, mo):
    """The state box of the reference's NLP branch (.../fnn/mpc_modeler_implementation_fnn.jl:146-153) in the SQP loop: rows of every
    iteration's QP.  Checked against the restatement with exact QP solves (same rows), and directly: the final trajectory is the
    network's own rollout, inside the box, with state rows active at the solution."""
    b, N, iters = 24, 20, 25
    f = mo.synthetic_fnn(act="tanh")
    n, m = 4, 2
    x_ref = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N + 1)); u_ref = np.tile(np.array([0.1, -0.2])[:, None], (1, N))
    Q, R, P, S = 100.0 * np.eye(n), 0.1 * np.eye(m), 150.0 * np.eye(n), np.zeros((m, m))
    umin, umax = -np.ones(m), np.ones(m)
    # the box: 0.6 of the range the unconstrained solutions of these instances sweep (tools: oracle sqp_fnn without rows), so that
    # the transients run into it; the initial states are moved inside
    xlo, xhi = np.array([-0.12, -0.58, -0.25, -0.35]), np.array([0.25, 0.09, 0.07, 0.22])
    X0 = x_ref[:, 0][None, :] + 0.5 * mo.splitmix_normal(0x5EED0005, 40, b, n)
    X0 = np.clip(X0, xlo + 0.02 * (xhi - xlo), xhi - 0.02 * (xhi - xlo))
    s = capi.Solver(n, m, N, b)
    s.sqp_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, S, P, umin, umax, act="tanh", xmin=xlo, xmax=xhi)
    s.sqp_fnn_start(X0)
    # three of these instances have an infeasible first QP (no elastic mode: the reference's Ipopt would report the NLP's
    # linearisation infeasible too, and JuMP.value would throw): they are skipped and named, the others are unaffected
    with pytest.raises(capi.AlmpcError) as ei:
        s.sqp_fnn_iterate(iters)
    assert ei.value.code == -6 and "infeasible" in str(ei.value)
    skipped = np.flatnonzero(s.sqp_fnn_skipped())
    r = s.get_results()
    s.close()
    infeasible = []
    for i in range(b):
        try:
            X, U, hist, na = mo.sqp_fnn(f, X0[i], x_ref, u_ref, Q, R, S, P, umin, umax, 1, x_min=xlo, x_max=xhi, return_active=True)
        except ValueError:
            infeasible.append(i)
    assert list(skipped) == infeasible and len(infeasible) == 3
    good = [i for i in range(b) if i not in infeasible]
    nact = 0
    for i in good:
        assert np.abs(r["x"][i] - mo.fnn_rollout(f, X0[i], r["u"][i])).max() <= 1e-5    # (instance 16 converges slowly: 2e-6 steps left)
        assert np.all(r["x"][i] <= xhi[:, None] + 1e-5) and np.all(r["x"][i] >= xlo[:, None] - 1e-5)
        nact += int(((r["x"][i][:, 1:] >= xhi[:, None] - 1e-6) | (r["x"][i][:, 1:] <= xlo[:, None] + 1e-6)).sum())
    assert nact > 10 * len(good), "the box never binds: test inputs too tame"
    for i in (0, 5, 11, 16, 23):
        X, U, hist, na = mo.sqp_fnn(f, X0[i], x_ref, u_ref, Q, R, S, P, umin, umax, iters, x_min=xlo, x_max=xhi, return_active=True)
        assert na >= 17
        assert np.abs(r["u"][i] - U).max() <= U_TOL and np.abs(r["x"][i] - X).max() <= 1e-5


def test_sqp_with_the_structured_qp_solver(capi, mo):
    """almpc_sqp_fnn_set_structured: every iteration's QP in its stage-wise form (k_riccati), no condensed design.  Same loop, same
    QPs: the iterates follow the restatement with stage-wise solves and end where the condensed path ends."""
    b, N, iters = 32, 50, 30
    f, s, kw, X0 = _setup(capi, mo, b, N)
    s.sqp_fnn_start(X0)
    st_c, de_c = s.sqp_fnn_iterate(iters)
    rc = s.get_results()
    s.close()
    n, m = 4, 2
    s2 = capi.Solver(n, m, N, b)
    s2.sqp_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, kw["x_ref"], kw["u_ref"], kw["Q"], kw["R"], kw["S"], kw["P"], kw["u_min"], kw["u_max"],
                     act="tanh", qp_solver="structured")
    s2.sqp_fnn_start(X0)
    st, de = s2.sqp_fnn_iterate(iters)
    r = s2.get_results()
    s2.close()
    assert np.all(r["status"] == 0)
    assert st[-1] <= 1e-7 and de[-1] <= 1e-12, (st, de)
    np.testing.assert_allclose(st[:6], st_c[:6], rtol=1e-5, atol=1e-8)     # same QPs, two solvers
    assert np.abs(r["u"] - rc["u"]).max() <= U_TOL and np.abs(r["x"] - rc["x"]).max() <= 1e-5
    for i in (0, 19):
        X, U, hist = mo.sqp_fnn(f, X0[i], kw["x_ref"], kw["u_ref"], kw["Q"], kw["R"], kw["S"], kw["P"], kw["u_min"], kw["u_max"], iters, structured=True)
        assert np.abs(r["u"][i] - U).max() <= U_TOL
    for i in range(b):
        assert np.abs(r["x"][i] - mo.fnn_rollout(f, X0[i], r["u"][i])).max() <= 1e-9
        assert mo.nlp_kkt_residual(f, X0[i], r["u"][i], kw["x_ref"], kw["u_ref"], kw["Q"], kw["R"], kw["S"], kw["P"], kw["u_min"], kw["u_max"]) <= 1e-5


def test_structured_qp_solver_with_state_box_and_rate_weight(capi, mo):
    """The stage-wise QP of the loop (k_sgains + k_sdual) with the rows the reference's NLP branch carries (state box,
    .../fnn/mpc_modeler_implementation_fnn.jl:146-153) and the input-rate weight (src/sub/design_mpc.jl:423-446): against the condensed
    route of the same loop (same QPs, other solver) and the restatement with exact QP solves."""
    b, N, iters = 24, 20, 25
    f = mo.synthetic_fnn(act="tanh")
    n, m = 4, 2
    x_ref = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N + 1)); u_ref = np.tile(np.array([0.1, -0.2])[:, None], (1, N))
    Q, R, P, S = 100.0 * np.eye(n), 0.1 * np.eye(m), 150.0 * np.eye(n), 0.2 * np.eye(m)
    umin, umax = -np.ones(m), np.ones(m)
    xlo, xhi = np.array([-0.12, -0.58, -0.25, -0.35]), np.array([0.25, 0.09, 0.07, 0.22])
    X0 = x_ref[:, 0][None, :] + 0.5 * mo.splitmix_normal(0x5EED0005, 40, b, n)
    X0 = np.clip(X0, xlo + 0.02 * (xhi - xlo), xhi - 0.02 * (xhi - xlo))
    res = {}
    for qp in ("condensed", "structured"):
        s = capi.Solver(n, m, N, b)
        s.sqp_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, S, P, umin, umax, act="tanh", xmin=xlo, xmax=xhi, qp_solver=qp)
        s.sqp_fnn_start(X0)
        try:
            s.sqp_fnn_iterate(iters)
            skipped = np.zeros(b, dtype=int)
        except capi.AlmpcError as e:      # instances whose first QP is infeasible are skipped and named (no elastic mode)
            assert e.code == -6
            skipped = s.sqp_fnn_skipped()
        res[qp] = (s.get_results(), np.flatnonzero(skipped))
        s.close()
    (rc, sc), (rs, ss) = res["condensed"], res["structured"]
    assert list(sc) == list(ss)
    good = [i for i in range(b) if i not in ss]
    assert len(good) >= b - 6
    assert np.abs(rs["u"][good] - rc["u"][good]).max() <= U_TOL and np.abs(rs["x"][good] - rc["x"][good]).max() <= 1e-4
    nact = 0
    for i in good:
        assert np.abs(rs["x"][i] - mo.fnn_rollout(f, X0[i], rs["u"][i])).max() <= 1e-5
        assert np.all(rs["x"][i] <= xhi[:, None] + 1e-5) and np.all(rs["x"][i] >= xlo[:, None] - 1e-5)
        nact += int(((rs["x"][i][:, 1:] >= xhi[:, None] - 1e-6) | (rs["x"][i][:, 1:] <= xlo[:, None] + 1e-6)).sum())
    assert nact > 5 * len(good), "the box never binds: test inputs too tame"
    for i in good[:3]:
        X, U, hist = mo.sqp_fnn(f, X0[i], x_ref, u_ref, Q, R, S, P, umin, umax, iters, x_min=xlo, x_max=xhi, structured="dual")
        assert np.abs(rs["u"][i] - U).max() <= U_TOL


def test_scaling_in_the_design_kernels_tail_equals_the_split_launches(capi, mo, monkeypatch):
    """An iteration's Jacobi scaling, scaled gradient and flag reset ride in the tail of k_design_ltv_reg (three launches less):
    the same operations on the same operands as k_design_scale / k_fs_scale behind it (ALMPC_DBG_SPLIT_SCALE) -- the same iterates."""
    res = {}
    for tag in ("fused", "split"):
        if tag == "split":
            monkeypatch.setenv("ALMPC_DBG_SPLIT_SCALE", "1")
        else:
            monkeypatch.delenv("ALMPC_DBG_SPLIT_SCALE", raising=False)
        f, s, kw, X0 = _setup(capi, mo, 32, 30)
        s.sqp_fnn_start(X0)
        s.sqp_fnn_iterate(12)
        res[tag] = s.get_results()
        s.close()
    monkeypatch.delenv("ALMPC_DBG_SPLIT_SCALE", raising=False)
    assert np.array_equal(res["fused"]["status"], res["split"]["status"])
    # (with the scaling in the tail, v0S_i = -G_i fS_i also comes out of the inverse's launch: another summation order, rounding level)
    assert np.abs(res["fused"]["u"] - res["split"]["u"]).max() <= 1e-9 and np.abs(res["fused"]["x"] - res["split"]["x"]).max() <= 1e-9


@pytest.mark.parametrize("env", ["ALMPC_DBG_SPLIT_PREPARE", "ALMPC_NO_GUESS_WS"])
def test_round5_folds_of_the_iteration_change_no_iterate(capi, mo, monkeypatch, env):
    """Round 5, two launches' worth of the iteration moved: (i) k_sqp_prepare runs as the head of k_design_ltv_reg
    (ALMPC_DBG_SPLIT_PREPARE=1: the separate launch) -- the same 256 threads doing the same arithmetic; (ii) the guess of the finish
    comes with the inverse of its working set, built by four waves per instance (k_guess_iterate_ws, two pivots per barrier) and
    installed by k_polish_sgl<1> (ALMPC_NO_GUESS_WS=1: k_guess_iterate, and the finish borders the rows beyond 32 in one at a time) --
    another order of operations for the same inverse.  N = 50: 100 inputs, about half of them on a bound after a few iterations.
    The merit safeguard is on in one of the runs (its test is part of the prepare step)."""
    res = {}
    for tag in ("new", "old"):
        if tag == "old":
            monkeypatch.setenv(env, "1")
        else:
            monkeypatch.delenv(env, raising=False)
        f, s, kw, X0 = _setup(capi, mo, 48, 50, amp=0.9)
        s.sqp_fnn_start(X0)
        s.sqp_fnn_iterate(6)
        st, de = s.sqp_fnn_iterate(6, step_rule="merit")
        res[tag] = (s.get_results(), st, de)
        s.close()
    monkeypatch.delenv(env, raising=False)
    a, b = res["new"][0], res["old"][0]
    assert np.array_equal(a["status"], b["status"]) and np.all(a["status"] == 0)
    assert np.abs(a["u"] - b["u"]).max() <= 1e-9 and np.abs(a["x"] - b["x"]).max() <= 1e-9
    assert np.allclose(res["new"][1], res["old"][1], rtol=1e-6, atol=1e-12)
    nb = (np.abs(a["u"]) >= 1.0 - 1e-12).reshape(48, -1).sum(axis=1)
    assert nb.max() > 32   # (working sets beyond the finish's register mode: the installed inverse was used)


@pytest.mark.parametrize("N", [33, 45, 64])
def test_guess_with_the_working_sets_inverse_at_the_edges_of_its_range(capi, mo, monkeypatch, N):
    """k_guess_iterate_ws serves 64 < nzs <= 128 and installs working sets of 33..64 rows.  N = 33: nz 66, sets around the lower edge
    (32 rows and fewer take the finish's register mode: the kernel must say "no start"); N = 45: nz 90, rows on both of its row waves;
    N = 64: nz 128, the largest shape, sets up to and beyond 64 rows (beyond: no start either).  Same iterates as k_guess_iterate
    (ALMPC_NO_GUESS_WS=1), and the exact restatement for one instance."""
    res = {}
    for tag in ("new", "old"):
        if tag == "old":
            monkeypatch.setenv("ALMPC_NO_GUESS_WS", "1")
        else:
            monkeypatch.delenv("ALMPC_NO_GUESS_WS", raising=False)
        f, s, kw, X0 = _setup(capi, mo, 24, N, amp=1.2)
        s.sqp_fnn_start(X0)
        s.sqp_fnn_iterate(10)
        res[tag] = s.get_results()
        s.close()
    monkeypatch.delenv("ALMPC_NO_GUESS_WS", raising=False)
    a, b = res["new"], res["old"]
    assert np.array_equal(a["status"], b["status"]) and np.all(a["status"] == 0)
    assert np.abs(a["u"] - b["u"]).max() <= 1e-9 and np.abs(a["x"] - b["x"]).max() <= 1e-9
    nb = (np.abs(a["u"]) >= 1.0 - 1e-12).reshape(24, -1).sum(axis=1)
    assert nb.max() > 32 and nb.min() < nb.max()
    X, U, hist = mo.sqp_fnn(f, X0[3], kw["x_ref"], kw["u_ref"], kw["Q"], kw["R"], kw["S"], kw["P"], kw["u_min"], kw["u_max"], 10)
    assert np.abs(a["u"][3] - U).max() <= U_TOL
