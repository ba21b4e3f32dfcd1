"""GPU tests of the structured (non-condensed) solve (second half): the multiple-shooting form the
reference builds (..linear.jl:48-60) solved stage by stage -- k_sdual (dual active set over affine Riccati sweeps, the handle's solver)
and behind it k_riccati (primal active set with Riccati-recursion subproblems, the safety net for saturated unstable plants).
Oracles: the exact condensed solver where the condensed problem is well conditioned, the numpy restatements of the two algorithms
(stagewise_oracle.solve_stage_dual: same decisions as k_sdual, so iteration counts agree; mpc_oracle.riccati_active_set), and
method-independent certificates for the shapes the condensed kernels cannot take (m*N > 128) or condition (unstable linearisations).
State rows, terminal equality and S on structured handles: tests/test_gpu_stagewise.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

U_TOL = 1e-6


@pytest.fixture(scope="module")
def so():
    import stagewise_oracle
    return stagewise_oracle


def _solve(capi, p, X0, **kw):
    s = capi.Solver(p.n, p.m, p.N, len(X0), structured=True, **kw)
    s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max)
    s.set_reference(p.x_ref, p.u_ref)
    s.update_initialization(X0)
    s.calculate()
    r = s.get_results()
    P = s.get_design()["P"]
    s.close()
    return r, P


def test_structured_solve_matches_the_exact_oracle_on_the_benchmark_plant(capi, mo, so):
    p = mo.quadrotor()
    X0 = np.concatenate([mo.quadrotor_x0_batch(40, a, first_instance=40 * k) for k, a in enumerate((0.3, 1.0, 3.0, 6.0))])
    r, P = _solve(capi, p, X0)
    assert np.all(r["status"] == 0)
    assert np.abs(P - p.P).max() <= 1e-9 * np.abs(p.P).max()
    for i in range(0, len(X0), 3):
        e = mo.solve_mpc_exact(p, X0[i])
        assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL and np.abs(r["x"][i] - e["x"]).max() <= 1e-5
    # same decisions as the restatement: iteration counts agree
    for i in (0, 50, 90, 130, 159):
        o = so.solve_mpc_stagewise(p, X0[i])
        assert o["status"] == 0 and o["iters"] == r["polish_iters"][i]
        assert np.abs(r["u"][i] - o["u"]).max() <= 1e-9
        assert np.abs(r["u"][i] - mo.solve_mpc_structured(p, X0[i])["u"]).max() <= 1e-8   # (the primal restatement: same optimum)
    np.testing.assert_allclose(r["e_u"], r["u"] - p.u_ref[None], atol=1e-14)
    np.testing.assert_allclose(r["e_x"], r["x"] - p.x_ref[None], atol=1e-12)
    np.testing.assert_array_equal(r["x"][:, :, 0], X0)


def test_structured_solve_takes_the_long_horizon_the_condensed_kernels_cannot(capi, mo, so):
    """Quadrotor, N = 50: m*N = 200 > 128 (almpc_create refuses a condensed handle).  Certificate: KKT conditions of the reference's
    own sparse statement cannot be formed cheaply per instance here, so: (i) the numpy restatement, (ii) the exact condensed oracle
    (numpy, no size limit) on a sample, (iii) dynamics, box, and first-order optimality of the condensed QP for every instance."""
    p = mo.quadrotor(N=50)
    with pytest.raises(capi.AlmpcError) as ei:
        capi.Solver(p.n, p.m, p.N, 4)
    assert ei.value.code == -4
    X0 = np.concatenate([mo.quadrotor_x0_batch(64, a, first_instance=64 * k) for k, a in enumerate((0.3, 1.0, 3.0))])
    r, _ = _solve(capi, p, X0)
    assert np.all(r["status"] == 0)
    ex, eu = r["e_x"], r["e_u"]
    pred = np.einsum("ij,bjk->bik", p.A, ex[:, :, :-1]) + np.einsum("ij,bjk->bik", p.B, eu)
    assert np.abs(pred - ex[:, :, 1:]).max() <= 1e-9 * max(1.0, np.abs(ex).max())
    assert np.all(r["u"] >= p.u_min[None, :, None]) and np.all(r["u"] <= p.u_max[None, :, None])
    _, _, H, F = mo.condense(p)
    V = eu.transpose(0, 2, 1).reshape(len(X0), -1)
    Gd = V @ H + X0 @ F.T                                        # gradient of the condensed QP at the structured solution
    lo = np.tile(p.u_min, p.N)[None] ; hi = np.tile(p.u_max, p.N)[None]
    sc = 1.0 / np.diag(H)[None]
    kkt = np.abs(V - np.clip(V - sc * Gd, lo, hi)).max(axis=1)   # projected-gradient residual, Jacobi scaled
    assert kkt.max() <= 1e-7
    for i in (0, 70, 140, 191):
        e = mo.solve_mpc_exact(p, X0[i])
        assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL
        o = so.solve_mpc_stagewise(p, X0[i])
        assert np.abs(r["u"][i] - o["u"]).max() <= 1e-9 and o["iters"] == r["polish_iters"][i]


def test_structured_fallback_solves_the_unstable_linearisations(capi, mo):
    """The relinearised Fnn batch of tests/test_gpu_batched_models.py: three of its 1024 linearisations are open-loop unstable
    (spectral radius up to 2.3, cond(H') 1e10 .. 5e16: the condensed Hessian is singular to working precision and the active-set
    finish runs into its cap).  With the structured fallback every instance of the batch is solved."""
    f = mo.synthetic_fnn()
    batch, N = 1024, 20
    x_ref, u_ref = np.array([0.2, -0.1, 0.05, 0.0]), np.array([0.1, -0.2])
    X0 = x_ref[None, :] + mo.splitmix_normal(0x5EED0004, 0, batch, 4) * 2.0
    A, B = capi.fnn_linearize(f.W_in, f.W_h, f.b_h, f.W_out, X0, np.repeat(u_ref[None], batch, 0), act=f.act)
    xr, ur = x_ref[:, None] * np.ones((4, N + 1)), u_ref[:, None] * np.ones((2, N))
    out = {}
    for fb in (False, True):
        s = capi.Solver(4, 2, N, batch, structured_fallback=fb)
        s.design_batched(A, B, 100.0 * np.eye(4), 0.1 * np.eye(2), None, None, [-1, -1], [1, 1])
        s.set_reference(xr, ur)
        s.update_initialization(X0)
        s.calculate()
        out[fb] = s.get_results()
        s.close()
    unstable = np.nonzero([np.max(np.abs(np.linalg.eigvals(A[i]))) > 1.5 for i in range(batch)])[0]   # 35 of the 1024, 5 / 464 / 785 among them
    assert len(unstable) >= 3
    unsolved = np.nonzero(out[False]["status"] != 0)[0]
    assert set(unsolved) <= set(unstable)           # what the condensed path alone leaves unsolved (a matter of rounding: up to all of them) ...
    assert np.all(out[True]["status"] == 0)         # ... is solved with the fallback: every instance of the batch has a certificate
    ok = out[False]["status"] == 0
    for key in ("u", "x", "polish_iters"):          # instances that were solved are not touched
        assert np.array_equal(out[False][key][ok], out[True][key][ok]), key
    for i in sorted(set(unsolved) | {5, 464, 785}):
        p = mo.make_problem(A[i], B[i], N, [-1, -1], [1, 1], x_ref=x_ref, u_ref=u_ref)
        o = mo.solve_mpc_structured(p, X0[i])
        assert o["status"] == 0 and np.abs(out[True]["u"][i] - o["u"]).max() <= U_TOL
        # method-independent: dynamics consistency and optimality through the adjoint multipliers' signs is what `o` certifies;
        # here additionally the cost is not worse than the exact condensed oracle's answer
        e = mo.solve_mpc_exact(p, X0[i])
        assert np.abs(out[True]["u"][i] - e["u"]).max() <= 1e-5


def test_structured_handle_with_per_instance_models_and_references(capi, mo):
    rng = np.random.default_rng(5)
    b, n, m, N = 24, 5, 3, 12
    As, Bs = [], []
    for _ in range(b):
        A = rng.standard_normal((n, n)); A *= (0.7 + 0.6 * rng.random()) / np.max(np.abs(np.linalg.eigvals(A)))
        As.append(A); Bs.append(rng.standard_normal((n, m)))
    As, Bs = np.stack(As), np.stack(Bs)
    X0 = 2.0 * rng.standard_normal((b, n))
    x_ref = 0.2 * np.ones((n, N + 1)); u_ref = np.tile(np.linspace(-0.2, 0.2, N), (m, 1))
    s = capi.Solver(n, m, N, b, structured=True)
    s.design_batched(As, Bs, 100.0 * np.eye(n), 0.1 * np.eye(m), None, None, -np.ones(m), np.ones(m))
    s.set_reference(x_ref, u_ref)
    s.update_initialization(X0)
    s.calculate()
    r = s.get_results()
    s.close()
    assert np.all(r["status"] == 0)
    for i in range(b):
        p = mo.make_problem(As[i], Bs[i], N, -np.ones(m), np.ones(m), x_ref=x_ref, u_ref=u_ref)
        e = mo.solve_mpc_exact(p, X0[i])
        assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL


def test_structured_handle_refuses_what_it_does_not_build(capi, mo):
    p = mo.double_integrator()
    s = capi.Solver(2, 1, 10, 2, structured=True)
    with pytest.raises(capi.AlmpcError) as ei:
        s.calculate()
    assert ei.value.code == -5
    # input-rate weight and state rows ARE built (k_sdual): tests/test_gpu_stagewise.py; here only that the designs are accepted
    s.design_shared(p.A, p.B, p.Q, p.R, 0.5 * np.eye(1), None, p.u_min, p.u_max)
    s.design_shared(p.A, p.B, p.Q, p.R, None, None, p.u_min, p.u_max, xmin=[-9, -9], xmax=[9, 9])
    s.design_shared(p.A, p.B, p.Q, p.R, None, None, p.u_min, p.u_max)
    s.update_initialization([[5.0, 0.0], [1.0, 0.0]])
    s.calculate()
    r = s.get_results()
    s.close()
    for i, x0 in enumerate(([5.0, 0.0], [1.0, 0.0])):
        assert np.abs(r["u"][i] - mo.solve_mpc_exact(p, np.array(x0))["u"]).max() <= U_TOL
    with pytest.raises(capi.AlmpcError):
        capi.Solver(40, 2, 10, 1, structured=True)      # n > 32
    # the re-linearisation pipeline is the stage-wise one on a structured handle (round 5: the test below); here only that it is taken
    f = mo.synthetic_fnn(act="tanh")
    s = capi.Solver(4, 2, 10, 2, structured=True)
    xr, ur = np.zeros((4, 11)), np.zeros((2, 10))
    s.relin_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, xr, ur, 100.0 * np.eye(4), 0.1 * np.eye(2), None, 150.0 * np.eye(4), [-1, -1], [1, 1], act="tanh")
    with pytest.raises(capi.AlmpcError) as ei:     # ... within the stage-wise solver's shape limits
        capi.Solver(30, 2, 160, 2, structured=True).relin_fnn_setup(np.zeros((4, 32)), [], [], np.zeros((30, 4)), None, None, 100.0 * np.eye(30),
                                                                    0.1 * np.eye(2), None, 150.0 * np.eye(30), [-1, -1], [1, 1], act="tanh")
    assert ei.value.code == -4
    # an SQP loop on a structured handle is the stage-wise one; it takes the input-rate weight too (tests/test_gpu_stagewise.py)
    s.sqp_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, xr, ur, 100.0 * np.eye(4), 0.1 * np.eye(2), 0.5 * np.eye(2), 150.0 * np.eye(4), [-1, -1], [1, 1], act="tanh")
    s.close()


def test_sqp_on_a_structured_handle_beyond_the_condensed_horizon(capi, mo):
    """ALMPC_FLAG_STRUCTURED + almpc_sqp_fnn_*: m N = 160 > 128 (a condensed handle cannot even be created), every iteration's QP in
    its stage-wise form, no condensed operand allocated.  Against the restatement's stage-wise loop and the NLP's own certificate."""
    n, m, N, b, iters = 4, 2, 80, 8, 25
    with pytest.raises(capi.AlmpcError):
        capi.Solver(n, m, N, b)
    f = mo.synthetic_fnn(act="tanh")
    x_ref = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N + 1)); u_ref = np.tile(np.array([0.1, -0.2])[:, None], (1, N))
    X0 = x_ref[:, 0][None, :] + 0.6 * mo.splitmix_normal(0x5EED0005, 0, b, n)
    Q, R, P = 100.0 * np.eye(n), 0.1 * np.eye(m), 150.0 * np.eye(n)
    S = np.zeros((m, m))
    umin, umax = -np.ones(m), np.ones(m)
    s = capi.Solver(n, m, N, b, structured=True)
    s.sqp_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, None, P, umin, umax, act="tanh")
    s.sqp_fnn_start(X0)
    st, de = s.sqp_fnn_iterate(iters, step_rule="merit")
    r = s.get_results()
    assert np.all(r["status"] == 0)
    assert st[-1] <= 1e-6 and de[-1] <= 1e-10, (st, de)
    for i in range(b):
        assert np.abs(r["x"][i] - mo.fnn_rollout(f, X0[i], r["u"][i])).max() <= 1e-9
        assert mo.nlp_kkt_residual(f, X0[i], r["u"][i], x_ref, u_ref, Q, R, S, P, umin, umax) <= 1e-5
    # a shared design afterwards takes the handle back to the plain structured solve
    p = mo.make_problem(np.eye(4) * 0.9, np.ones((4, 2)) * 0.1, N, umin, umax)
    s.design_shared(p.A, p.B, p.Q, p.R, None, None, p.u_min, p.u_max)
    s.update_initialization(X0)
    s.calculate()
    r2 = s.get_results()
    assert np.abs(r2["u"][0] - mo.solve_mpc_structured(p, X0[0])["u"]).max() <= U_TOL
    s.close()


def test_sqp_loop_solves_an_indefinite_condensed_qp_through_the_stage_wise_form(capi, mo):
    """An Fnn with identity activation and spectral radius 1.6 over N = 50 stages: the model is linear, so the NLP of the reference's
    NonLinearProgramming branch is a convex QP -- but its condensed Hessian (cond ~ 1.6^100) is indefinite to working precision, the
    design kernels flag every instance and the SQP loop can only contain them (ALMPC_ERR_NUMERIC, iterate kept).  With the structured
    fallback the loop solves each iteration's QP in its stage-wise form (k_riccati with time-varying stage models, defects, state
    errors and the input gradient) and lands on the optimum of the linear problem in one iteration."""
    n, m, N, b = 4, 2, 50, 6
    f = mo.synthetic_fnn(act="identity")
    f.b_h = [0.0 * v for v in f.b_h]
    A0, B0 = f.jacobian(np.zeros(n), np.zeros(m))
    f.W_out = f.W_out * (1.6 / np.max(np.abs(np.linalg.eigvals(A0))))
    A, B = f.jacobian(np.zeros(n), np.zeros(m))
    x_ref, u_ref = np.zeros((n, N + 1)), np.zeros((m, N))
    Q, R, P = 100.0 * np.eye(n), 0.1 * np.eye(m), 150.0 * np.eye(n)
    X0 = 0.25 * mo.splitmix_normal(0x5EED0009, 0, b, n)
    out = {}
    for fb in (False, True):
        s = capi.Solver(n, m, N, b, structured_fallback=fb)   # (False: switched off explicitly -- the library default is on)
        s.sqp_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, None, P, -np.ones(m), np.ones(m), act="identity")
        s.sqp_fnn_start(X0)
        if fb:
            st, de = s.sqp_fnn_iterate(3)
            assert s.sqp_fnn_skipped().sum() == 0
            out["hist"] = (st, de)
        else:
            with pytest.raises(capi.AlmpcError) as ei:
                s.sqp_fnn_iterate(3)
            assert ei.value.code == -6 and s.sqp_fnn_skipped().sum() >= 1
        out[fb] = s.get_results(want=("u", "x", "status"))
        s.close()
    st, de = out["hist"]
    assert st[0] > 1e-3 and st[-1] <= 1e-9 and de[-1] <= 1e-9, (st, de)          # linear model: converged after the first iteration
    p = mo.make_problem(A, B, N, -np.ones(m), np.ones(m), P=P)
    for i in range(b):
        o = mo.solve_mpc_structured(p, X0[i])
        assert o["status"] == 0 and np.abs(out[True]["u"][i] - o["u"]).max() <= 1e-6
    assert ((np.abs(out[True]["u"]) >= 1.0).sum()) >= 1                             # the box is active somewhere


def test_horizon_continuation_and_warm_start(capi, mo):
    """almpc_set_start_from: the condensed path at N = 30 hands the stage-wise path at N = 50 its working set.  Same optimum as
    the plain structured solve (the start does not matter), far fewer working-set changes; and opts.warm_start = 1 on a structured
    handle (previous inputs shifted by one stage) along a closed loop."""
    b = 192
    amp = np.array([0.3, 1.0, 3.0])[np.arange(b) % 3]
    X0 = mo.splitmix_normal(0x5EED0002, 0, b, 12) * mo.QUADROTOR_X0_SCALE[None] * amp[:, None]
    X0[0] = mo.splitmix_normal(0x5EED0002, 1871, 1, 12)[0] * mo.QUADROTOR_X0_SCALE * 3.0   # the benchmark batch's hardest instance
    p30, p50 = mo.quadrotor(30), mo.quadrotor(50)
    s50 = capi.Solver(12, 4, 50, b, structured=True)
    s50.design_shared(p50.A, p50.B, p50.Q, p50.R, None, None, p50.u_min, p50.u_max)
    s50.set_reference(p50.x_ref, p50.u_ref)
    s50.update_initialization(X0)
    s50.calculate()
    plain = s50.get_results()
    s30 = capi.Solver(12, 4, 30, b)
    s30.design_shared(p30.A, p30.B, p30.Q, p30.R, None, None, p30.u_min, p30.u_max)
    s30.set_reference(p30.x_ref, p30.u_ref)
    s30.update_initialization(X0)
    s30.calculate(sync=False)
    s50.start_from(s30)          # enqueued behind s30's step on s50's stream: no host wait in between
    s50.calculate()
    cont = s50.get_results()
    assert np.all(plain["status"] == 0) and np.all(cont["status"] == 0)
    assert np.abs(cont["u"] - plain["u"]).max() <= 1e-8 and np.abs(cont["x"] - plain["x"]).max() <= 1e-7
    # (iterations of the dual method count the start's purged rows as well: a continued start is a handful of corrections)
    assert cont["polish_iters"].max() < plain["polish_iters"].max() and cont["polish_iters"].mean() <= plain["polish_iters"].mean(), \
        (cont["polish_iters"].max(), plain["polish_iters"].max())
    for i in (0, 7, 100):
        assert np.abs(cont["u"][i] - mo.solve_mpc_structured(p50, X0[i])["u"]).max() <= U_TOL
    # the start is consumed by one step: the next plain step starts from the clipped LQR point again
    s50.calculate()
    again = s50.get_results()
    assert np.array_equal(again["polish_iters"], plain["polish_iters"])
    # a source with a longer horizon, another batch or another shape is refused
    with pytest.raises(capi.AlmpcError):
        s30.start_from(s50)
    # closed loop with warm starts: x+ = A x + B u[:,1]; every step equals the cold solve of the same state
    x = X0.copy()
    warm = capi.default_opts(warm_start=1)
    s50.update_initialization(x); s50.calculate()
    for _ in range(3):
        u0 = s50.get_results(want=("u",))["u"][:, :, 0]
        x = x @ p50.A.T + u0 @ p50.B.T
        s50.update_initialization(x)
        s50.calculate(warm)
        rw = s50.get_results()
        s50.calculate()
        rc = s50.get_results()
        assert np.all(rw["status"] == 0)
        assert np.abs(rw["u"] - rc["u"]).max() <= 1e-8
        # (a dual method gains less from a guessed working set than a primal one: the guess's rows cost one sweep instead of two, the
        # wrong ones are purged, the missing ones added as from a cold start)
        assert rw["polish_iters"].sum() <= 1.2 * rc["polish_iters"].sum()
        s50.update_initialization(x); s50.calculate(warm)   # leave the warm result in place for the next shift
    s30.close(); s50.close()


def test_relinearisation_pipeline_on_a_structured_handle_beyond_the_condensed_horizon(capi, mo):
    """Round-4 review, item 5: the reference's Fnn-LP delegation has no horizon limit (.../fnn/mpc_modeler_implementation_fnn.jl:23-58);
    almpc_relin_fnn_* refused ALMPC_FLAG_STRUCTURED handles.  A quadrotor-size network (n 12, m 4) at N = 50 -- m N = 200, no condensed
    handle exists -- re-linearised at every instance's own state: k_fnn_jacobian_w -> k_sgains -> k_sdual on the device, every
    instance against the exact oracle on ITS OWN (A_i, B_i), cold and in closed loop with warm steps, with a state box and S, and as
    a group of two handles."""
    n, m, N, batch = 12, 4, 50, 256
    f = mo.synthetic_fnn(n=n, m=m, H=24, L=2, seed=0x5EED0044, act="tanh")
    x_ref, u_ref = np.zeros(n), np.zeros(m)
    X0 = 0.5 * mo.splitmix_normal(0x5EED0045, 0, batch, n)
    Q, R = 100.0 * np.eye(n), 0.1 * np.eye(m)
    umin, umax = -0.3 * np.ones(m), 0.3 * np.ones(m)
    Al, Bl = f.jacobian(x_ref, u_ref)
    P = mo.dare(Al, Bl, Q, R)     # terminal weight as the reference takes it: linearisation at the last reference
    xr, ur = np.tile(x_ref[:, None], (1, N + 1)), np.tile(u_ref[:, None], (1, N))
    sv = capi.Solver(n, m, N, batch, structured=True)
    sv.relin_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, xr, ur, Q, R, None, P, umin, umax, act=f.act)
    sv.update_initialization(X0)
    sv.relin_fnn_step()
    r = sv.get_results()
    assert np.all(r["status"] == 0), np.bincount(r["status"])
    nact = 0
    for i in range(0, batch, 8):
        Ai, Bi = f.jacobian(X0[i], u_ref)
        p = mo.make_problem(Ai, Bi, N, umin, umax, x_ref=x_ref, u_ref=u_ref, P=P)
        e = mo.solve_mpc_structured(p, X0[i])
        assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL, i
        nact += int(((e["u"] <= umin[:, None] + 1e-12) | (e["u"] >= umax[:, None] - 1e-12)).sum())
    assert nact > 40
    # closed loop on the network with warm steps: every warm step equals the cold solve of the same state
    warm = capi.default_opts(warm_start=1)
    cold = capi.Solver(n, m, N, batch, structured=True)
    cold.relin_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, xr, ur, Q, R, None, P, umin, umax, act=f.act)
    for _ in range(3):
        sv.relin_fnn_advance()
        sv.relin_fnn_step(warm)
        rw = sv.get_results(want=("u", "x", "status"))
        cold.update_initialization(rw["x"][:, :, 0].copy())
        cold.relin_fnn_step()
        rc = cold.get_results(want=("u", "status"))
        assert np.all(rw["status"] == 0) and np.all(rc["status"] == 0)
        assert np.abs(rw["u"] - rc["u"]).max() <= 1e-6
    cold.close()
    sv.close()
    # state box + input-rate weight + a horizon-varying input reference (the S terms of the reference reach the solver's base vector)
    urv = 0.02 * np.sin(np.arange(N))[None, :] * np.ones((m, 1))
    xbox = 2.0 * np.ones(n)
    S = 3.0 * np.eye(m)
    sb = capi.Solver(n, m, N, batch, structured=True)
    sb.relin_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, xr, urv, Q, R, S, P, umin, umax, act=f.act, xmin=-xbox, xmax=xbox)
    sb.update_initialization(X0)
    sb.relin_fnn_step()
    rb = sb.get_results(want=("u", "status"))
    sb.close()
    checked = 0
    for i in range(0, batch, 16):
        Ai, Bi = f.jacobian(X0[i], urv[:, 0])
        p = mo.make_problem(Ai, Bi, N, umin, umax, x_ref=xr, u_ref=urv, P=P, s=3.0, x_min=-xbox, x_max=xbox)
        try:
            e = mo.solve_mpc_exact(p, X0[i])
        except ValueError:
            assert rb["status"][i] == 3, i
            continue
        assert rb["status"][i] == 0, (i, rb["status"][i])
        assert np.abs(rb["u"][i] - e["u"]).max() <= 1e-5, i
        checked += 1
    assert checked >= 8
    # a group of two handles on device 0 equals the single handle
    g = capi.Group(n, m, N, batch, [0, 0], structured=True)
    g.relin_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, xr, ur, Q, R, None, P, umin, umax, act=f.act)
    g.update_initialization(X0)
    g.relin_fnn_step()
    rg = g.get_results(want=("u", "status"))
    g.close()
    assert np.array_equal(rg["status"], r["status"]) and np.array_equal(rg["u"], r["u"])
