"""GPU parity tests for per-instance models (almpc_design_batched): every instance has its own (A_i, B_i).  The oracle builds
one MPCProblem per instance (the reference's QP for given (A, B)) and solves it exactly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

U_TOL = 1e-5   # BASELINE.json: |u - u*|_inf <= 1e-5
X_TOL = 1e-4


def quad_family(mo, batch, seed=7, spread=0.25):
    """Quadrotors with per-instance mass and inertia (within +-spread of the nominal) -> per-instance (A_i, B_i)."""
    rng = np.random.default_rng(seed)
    As, Bs = [], []
    for _ in range(batch):
        f = 1.0 + spread * (2.0 * rng.random(4) - 1.0)
        A, B = mo.quadrotor_model(mass=0.5 * f[0], J=(4e-3 * f[1], 4e-3 * f[2], 8e-3 * f[3]))
        As.append(A); Bs.append(B)
    return np.stack(As), np.stack(Bs)


def random_family(n, m, batch, seed):
    rng = np.random.default_rng(seed)
    As, Bs = [], []
    for _ in range(batch):
        A = rng.standard_normal((n, n))
        A *= (0.6 + 0.5 * rng.random()) / max(1e-9, np.max(np.abs(np.linalg.eigvals(A))))  # spectral radius 0.6 .. 1.1
        As.append(A); Bs.append(rng.standard_normal((n, m)))
    return np.stack(As), np.stack(Bs)


def solve_batched(capi, As, Bs, N, umin, umax, X0, opts=None, x_ref=None, u_ref=None, q=100.0, r=0.1, s=0.0, P=None, **design_kw):
    b, n, m = As.shape[0], As.shape[1], Bs.shape[2]
    sv = capi.Solver(n, m, N, b)
    sv.design_batched(As, Bs, q * np.eye(n), r * np.eye(m), s * np.eye(m), P, umin, umax, **design_kw)
    if x_ref is not None:
        sv.set_reference(x_ref, u_ref)
    sv.update_initialization(X0)
    sv.calculate(opts)
    res = sv.get_results()
    return sv, res


def test_design_batched_matches_oracle(capi, mo):
    As, Bs = quad_family(mo, 6)
    sv = capi.Solver(12, 4, 30, 6)
    sv.design_batched(As, Bs, 100 * np.eye(12), 0.1 * np.eye(4), None, None, [-2, -.05, -.05, -.02], [3, .05, .05, .02])
    for i in range(6):
        p = mo.make_problem(As[i], Bs[i], 30, [-2, -.05, -.05, -.02], [3, .05, .05, .02])
        _, _, H, F = mo.condense(p)
        g = sv.get_design_instance(i)
        assert np.abs(g["H"] - H).max() <= 1e-11 * np.abs(H).max()
        assert np.abs(g["F"] - F).max() <= 1e-11 * np.abs(F).max()
        assert np.abs(g["d"] - mo.jacobi_scaling(H)).max() <= 1e-12 * np.abs(g["d"]).max()
    sv.close()


@pytest.mark.parametrize("amp", [0.5, 2.0])
def test_quadrotor_family_vs_exact_oracle(capi, mo, amp):
    b = 40
    As, Bs = quad_family(mo, b)
    X0 = mo.quadrotor_x0_batch(b, amp, first_instance=5000)
    umin, umax = [-2, -.05, -.05, -.02], [3, .05, .05, .02]
    sv, r = solve_batched(capi, As, Bs, 30, umin, umax, X0)
    sv.close()
    assert np.all(r["status"] == 0)
    nact = 0
    for i in range(b):
        p = mo.make_problem(As[i], Bs[i], 30, umin, umax)
        e = mo.solve_mpc_exact(p, X0[i])
        assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL
        assert np.abs(r["x"][i] - e["x"]).max() <= X_TOL
        nact += (np.isclose(e["u"], p.u_min[:, None]) | np.isclose(e["u"], p.u_max[:, None])).sum()
    if amp >= 2.0:
        assert nact > 50, "test input leaves the bounds inactive"


def test_admm_iterate_matches_oracle_per_instance(capi, mo):
    """polish = 0: the ADMM iterate of k_admm_inst (LDS-resident KKT inverse, column walk) is the oracle's iterate."""
    b = 12
    As, Bs = quad_family(mo, b, seed=11)
    X0 = mo.quadrotor_x0_batch(b, 1.5, first_instance=77)
    umin, umax = [-2, -.05, -.05, -.02], [3, .05, .05, .02]
    for kw, prof in ((dict(rho=0.1), "scalar"), (dict(rho=30.0), "stiffness")):
        sv, r = solve_batched(capi, As, Bs, 30, umin, umax, X0, capi.default_opts(max_iter=10, check_every=10, polish=0, **kw),
                              rho_profile=prof, **kw)
        sv.close()
        for i in range(b):
            p = mo.make_problem(As[i], Bs[i], 30, umin, umax)
            des = mo.design_shared(p, rho=kw["rho"], rho_profile=prof)
            fs = des["Fs"] @ (X0[i] - p.x_ref[:, 0]) + des["fS"]
            a = mo.admm_box(des["Hs"], fs, des["lo"], des["hi"], rho=des["rho_vec"], sigma=des["sigma"], max_iter=10, check_every=10,
                            Minv=des["Minv"], unscale=des["d"])
            v = np.clip(a["z"] * des["d"], des["lo"] * des["d"], des["hi"] * des["d"])
            assert r["iters"][i] == a["iters"] and r["status"][i] == a["status"]
            assert np.abs(r["e_u"][i].T.reshape(-1) - v).max() <= 1e-9 * max(1.0, np.abs(v).max())


# (the second row: nz on both sides of every size class of the register inverses -- 16 / 32 / 48 / 64 columns of the one-wave kernel,
# 65 .. 128 for the multi-wave one)
@pytest.mark.parametrize("n,m,N", [(1, 1, 1), (2, 1, 12), (3, 2, 7), (5, 3, 9), (4, 2, 20), (7, 5, 25), (2, 1, 128), (16, 8, 16),
                                   (2, 1, 16), (2, 1, 17), (2, 2, 16), (3, 3, 11), (2, 3, 16), (2, 7, 7), (3, 1, 63), (2, 4, 16),
                                   (3, 5, 13), (2, 1, 97), (3, 1, 127)])
def test_random_plant_families_of_many_shapes(capi, mo, n, m, N):
    b = 17
    As, Bs = random_family(n, m, b, seed=1000 * n + 10 * m + N)
    rng = np.random.default_rng(5)
    X0 = rng.standard_normal((b, n)) * 2.0
    umin, umax = -np.ones(m), np.ones(m)
    sv, r = solve_batched(capi, As, Bs, N, umin, umax, X0)
    sv.close()
    for i in range(b):
        p = mo.make_problem(As[i], Bs[i], N, umin, umax)
        e = mo.solve_mpc_exact(p, X0[i])
        if r["status"][i] == 0:
            assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL * max(1.0, np.abs(e["x"]).max())
    assert np.all(r["status"] == 0), np.bincount(r["status"], minlength=4)   # every instance solved (stage-wise redo of what the condensed finish leaves: default on)


def test_more_than_256_designs_beyond_64_variables(capi, mo):
    """nz = 65 with 300 models: the throughput variant of the multi-wave inverse (more matrices than CUs), odd sizes in both halves."""
    n, m, N, b = 3, 5, 13, 300
    As, Bs = random_family(n, m, b, seed=4242)
    X0 = np.random.default_rng(6).standard_normal((b, n)) * 2.0
    umin, umax = -np.ones(m), np.ones(m)
    sv, r = solve_batched(capi, As, Bs, N, umin, umax, X0)
    sv.close()
    assert np.all(r["status"] == 0), np.bincount(r["status"], minlength=4)
    for i in range(0, b, 37):
        if r["status"][i] == 0:
            e = mo.solve_mpc_exact(mo.make_problem(As[i], Bs[i], N, umin, umax), X0[i])
            assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL * max(1.0, np.abs(e["x"]).max())


def test_identical_models_reproduce_the_shared_path(capi, mo):
    p = mo.quadrotor()
    b = 48
    X0 = mo.quadrotor_x0_batch(b, 2.0, first_instance=321)
    As, Bs = np.repeat(p.A[None], b, 0), np.repeat(p.B[None], b, 0)
    sv, r = solve_batched(capi, As, Bs, 30, p.u_min, p.u_max, X0, P=p.P)
    # same handle back on the shared-model path
    sv.design_shared(p.A, p.B, p.Q, p.R, p.S, p.P, p.u_min, p.u_max)
    sv.set_reference(p.x_ref, p.u_ref)
    sv.update_initialization(X0)
    sv.calculate()
    s = sv.get_results()
    sv.close()
    assert np.array_equal(r["status"], s["status"])
    assert np.abs(r["u"] - s["u"]).max() <= 1e-9 and np.abs(r["x"] - s["x"]).max() <= 1e-7


def test_references_rate_weight_and_terminal_weight_options(capi, mo):
    """Non-zero references, S != 0 (f_S term scaled per instance), shared P and per-instance P."""
    b, n, m, N = 9, 4, 2, 15
    As, Bs = random_family(n, m, b, seed=3)
    As *= 0.9
    rng = np.random.default_rng(1)
    X0 = rng.standard_normal((b, n))
    x_ref = 0.3 * np.ones((n, N + 1)); u_ref = np.tile(np.linspace(-0.3, 0.3, N), (m, 1))
    umin, umax = -np.ones(m), np.ones(m)
    Pshared = 50.0 * np.eye(n)
    for P in (None, Pshared, np.stack([(40.0 + i) * np.eye(n) for i in range(b)])):
        sv, r = solve_batched(capi, As, Bs, N, umin, umax, X0, x_ref=x_ref, u_ref=u_ref, s=0.7, P=P)
        sv.close()
        for i in range(b):
            Pi = None if P is None else (P if P.ndim == 2 else P[i])
            p = mo.make_problem(As[i], Bs[i], N, umin, umax, x_ref=x_ref, u_ref=u_ref, s=0.7, P=Pi)
            e = mo.solve_mpc_exact(p, X0[i])
            assert r["status"][i] == 0 and np.abs(r["u"][i] - e["u"]).max() <= U_TOL


def test_config4_relinearised_fnn_per_instance(capi, mo):
    """BASELINE configs[3] as an extension of the reference: the Fnn model is linearised at every instance's own
    (x0_i, u_ref) by the Jacobian kernel, and each instance gets the reference's QP for its (A_i, B_i); N = 20, batch = 1024."""
    f = mo.synthetic_fnn()
    batch, N = 1024, 20
    x_ref, u_ref = np.array([0.2, -0.1, 0.05, 0.0]), np.array([0.1, -0.2])
    X0 = x_ref[None, :] + mo.splitmix_normal(0x5EED0004, 0, batch, 4) * 2.0
    A, B = capi.fnn_linearize(f.W_in, f.W_h, f.b_h, f.W_out, X0, np.repeat(u_ref[None], batch, 0), act=f.act)
    # (three of these linearisations are open-loop unstable, cond(H') up to 5e16: the condensed path leaves them unsolved and the
    # structured fallback -- k_riccati on the multiple-shooting form -- solves them: tests/test_gpu_structured.py has the details)
    sv = capi.Solver(4, 2, N, batch, structured_fallback=True)
    sv.design_batched(A, B, 100.0 * np.eye(4), 0.1 * np.eye(2), None, None, [-1, -1], [1, 1])
    sv.set_reference(x_ref[:, None] * np.ones((4, N + 1)), u_ref[:, None] * np.ones((2, N)))
    sv.update_initialization(X0)
    sv.calculate()
    r = sv.get_results()
    sv.close()
    assert np.all(r["status"] == 0)
    nact = 0
    for i in range(0, batch, 16):
        Ai, Bi = f.jacobian(X0[i], u_ref)
        p = mo.make_problem(Ai, Bi, N, [-1, -1], [1, 1], x_ref=x_ref, u_ref=u_ref)
        e = mo.solve_mpc_exact(p, X0[i])
        if r["status"][i] == 0:
            assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL
            # x is the rollout of u through (A_i, B_i): compared directly where the linearisation is stable; an unstable one
            # amplifies a 1e-7 difference in u by rho(A_i)^N, so there the rollout identity is checked on the GPU's own u
            if np.max(np.abs(np.linalg.eigvals(Ai))) <= 1.0:
                assert np.abs(r["x"][i] - e["x"]).max() <= X_TOL
            ex = mo.rollout(p, X0[i], (r["u"][i] - p.u_ref).T.reshape(-1))["x"]
            assert np.abs(r["x"][i] - ex).max() <= 1e-9 * max(1.0, np.abs(ex).max())
        nact += ((e["u"] <= -1) | (e["u"] >= 1)).sum()
    assert nact > 50


def test_batched_error_behaviour(capi, mo):
    p = mo.double_integrator()
    sv = capi.Solver(2, 1, 10, 4)
    A, B = np.repeat(p.A[None], 4, 0), np.repeat(p.B[None], 4, 0)
    with pytest.raises(capi.AlmpcError) as ei:   # get_design_instance before a batched design
        sv.get_design_instance(0)
    assert ei.value.code == -5
    sv.design_batched(A, B, p.Q, p.R, None, None, p.u_min, p.u_max)
    with pytest.raises(capi.AlmpcError) as ei:
        sv.advance_plant()
    assert ei.value.code == -4
    with pytest.raises(capi.AlmpcError) as ei:
        sv.get_design_instance(4)
    assert ei.value.code == -1
    Abad = A.copy(); Abad[2, 0, 0] = np.nan   # a broken model: the numeric error names the instance
    with pytest.raises(capi.AlmpcError) as ei:
        sv.design_batched(Abad, B, p.Q, p.R, None, None, p.u_min, p.u_max)
    assert ei.value.code == -6 and "instance 2" in str(ei.value)
    with pytest.raises(capi.AlmpcError) as ei:   # the same with a given P: caught by the device-side pivot checks
        sv.design_batched(Abad, B, p.Q, p.R, None, p.P, p.u_min, p.u_max)
    assert ei.value.code == -6 and "instance 2" in str(ei.value)
    sv.close()


def test_mirror_relinearises_every_step(pkg, capi, mo):
    """Host mirror, kw mpc_linearization='step': update_initialization! re-linearises the Fnn model at every instance's own state,
    calculate! solves each instance's own QP.  The default ('reference') stays the reference's single linearisation."""
    f = mo.synthetic_fnn()
    sys_ = pkg.ConstrainedBlackBoxControlDiscreteSystem(pkg.Fnn(f.W_in, f.W_h, f.b_h, f.W_out, f.act), 4, 2,
                                                        pkg.Hyperrectangle([-10] * 4, [10] * 4), pkg.Hyperrectangle([-1, -1], [1, 1]))
    x_ref, u_ref = [0.2, -0.1, 0.05, 0.0], [0.1, -0.2]
    batch, N = 64, 20
    C = pkg.proceed_controller(sys_, "model_predictive_control", N, 1, x_ref, u_ref, mpc_batch=batch, mpc_linearization="step")
    P = C.tuning.terminal_ingredient.P
    X0 = np.asarray(x_ref)[None, :] + mo.splitmix_normal(0x5EED0004, 100, batch, 4)
    for step in range(2):   # two consecutive steps from different states: a new design each time
        res = pkg._model_predictive_control_computation(C, X0 * (1.0 + 0.5 * step))
        st = C.tuning.modeler.last_status
        for i in range(0, batch, 5):
            Ai, Bi = f.jacobian(X0[i] * (1.0 + 0.5 * step), np.asarray(u_ref))
            p = mo.make_problem(Ai, Bi, N, [-1, -1], [1, 1], x_ref=np.asarray(x_ref), u_ref=np.asarray(u_ref), P=P)
            e = mo.solve_mpc_exact(p, X0[i] * (1.0 + 0.5 * step))
            if st[i] == 0:
                assert np.abs(res.u[i] - e["u"]).max() <= U_TOL
        assert np.all(st == 0), np.bincount(st, minlength=4)
    C.tuning.modeler.solver.close()
    with pytest.raises(ValueError):
        pkg.proceed_controller(sys_, "model_predictive_control", N, 1, x_ref, u_ref, mpc_linearization="sometimes")


def test_relin_pipeline_on_device_equals_host_path(capi, mo):
    """almpc_relin_fnn_* (BASELINE configs[3]: Jacobians -> per-instance designs -> step, all on the handle's stream, no host pointers)
    gives bit for bit what the host-pointer route gives (almpc_fnn_linearize -> almpc_design_batched -> almpc_calculate), for a
    time-varying input reference with an input-rate weight too (the reference-dependent vectors are re-made on the device)."""
    f = mo.synthetic_fnn()
    batch, N, n, m = 200, 20, 4, 2
    x_ref = np.array([0.2, -0.1, 0.05, 0.0])[:, None] * np.ones((n, N + 1))
    for S, u_ref in ((None, np.array([0.1, -0.2])[:, None] * np.ones((m, N))),
                     (0.3 * np.eye(m), np.stack([np.linspace(0.1, -0.1, N), np.linspace(-0.2, 0.2, N)]))):
        X0 = x_ref[:, 0][None, :] + mo.splitmix_normal(0x5EED0004, 7, batch, n)
        Q, R = 100.0 * np.eye(n), 0.1 * np.eye(m)
        Al, Bl = capi.fnn_linearize(f.W_in, f.W_h, f.b_h, f.W_out, x_ref[:, -1][None], u_ref[:, -1][None], act=f.act)
        P = capi.dare(Al[0], Bl[0], Q, R)
        opts = capi.default_opts()
        # host-pointer route
        A, B = capi.fnn_linearize(f.W_in, f.W_h, f.b_h, f.W_out, X0, np.repeat(u_ref[:, 0][None], batch, 0), act=f.act)
        sh = capi.Solver(n, m, N, batch)
        sh.design_batched(A, B, Q, R, S, P, [-1, -1], [1, 1])
        sh.set_reference(x_ref, u_ref)
        sh.update_initialization(X0)
        sh.calculate(opts)
        a = sh.get_results()
        sh.close()
        # device-resident pipeline, two steps (the second from other states: new Jacobians, new designs)
        sd = capi.Solver(n, m, N, batch, timing=True)
        sd.relin_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, S, P, [-1, -1], [1, 1], act=f.act)
        sd.update_initialization(0.5 * X0)
        sd.relin_fnn_step(opts)
        sd.update_initialization(X0)
        sd.relin_fnn_step(opts)
        b = sd.get_results()
        t = sd.relin_fnn_timing()
        sd.close()
        for key in ("status", "iters", "polish_iters", "u", "x", "e_u", "e_x"):
            assert np.array_equal(a[key], b[key]), key
        assert np.all(a["status"] == 0) and t["design_ms"] > 0 and t["step_ms"] > 0
        if S is None:
            # without the stage-wise redo (switched off explicitly) the condensed path gives up on the open-loop unstable linearisations;
            # with it (the default, above) nothing stays unsolved and the others keep their results bit for bit
            sn = capi.Solver(n, m, N, batch, structured_fallback=False)
            sn.relin_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, None, P, [-1, -1], [1, 1], act=f.act)
            sn.update_initialization(X0)
            sn.relin_fnn_step(opts)
            c = sn.get_results()
            sn.close()
            ok = c["status"] == 0
            assert ok.mean() >= 0.9
            assert np.array_equal(a["u"][ok], c["u"][ok])
            for i in np.nonzero(~ok)[0][:4]:
                pi = mo.make_problem(A[i], B[i], N, [-1, -1], [1, 1], x_ref=x_ref, u_ref=u_ref, P=P)
                ex = mo.rollout(pi, X0[i], (a["u"][i] - pi.u_ref).T.reshape(-1))["x"]
                assert np.abs(a["x"][i] - ex).max() <= 1e-9 * max(1.0, np.abs(ex).max())
        for i in range(0, batch, 23):
            if a["status"][i] == 0:
                p = mo.make_problem(A[i], B[i], N, [-1, -1], [1, 1], x_ref=x_ref, u_ref=u_ref, s=0.0 if S is None else 0.3, P=P)
                assert np.abs(b["u"][i] - mo.solve_mpc_exact(p, X0[i])["u"]).max() <= U_TOL


def test_relin_closed_loop_warm_steps(capi, mo):
    """Closed loop of the black-box model itself: step -> almpc_relin_fnn_advance (x0 <- fnn(x0, u[:,1]) on the device) -> step.
    A warm step (opts.warm_start = 1: working-set guess = previous inputs shifted one stage, no ADMM phase, one inverse per design)
    reaches the same optimum as a cold one, and both match the exact oracle on the instance's own linearisation."""
    f = mo.synthetic_fnn()
    batch, N, n, m = 256, 20, 4, 2
    x_ref = np.array([0.2, -0.1, 0.05, 0.0])[:, None] * np.ones((n, N + 1))
    u_ref = np.array([0.1, -0.2])[:, None] * np.ones((m, N))
    Q, R = 100.0 * np.eye(n), 0.1 * np.eye(m)
    Al, Bl = capi.fnn_linearize(f.W_in, f.W_h, f.b_h, f.W_out, x_ref[:, -1][None], u_ref[:, -1][None], act=f.act)
    P = capi.dare(Al[0], Bl[0], Q, R)
    X0 = x_ref[:, 0][None, :] + mo.splitmix_normal(0x5EED0004, 11, batch, n) * 1.5
    sw = capi.Solver(n, m, N, batch, timing=True, structured_fallback=True)
    sc = capi.Solver(n, m, N, batch, timing=True, structured_fallback=True)
    for s in (sw, sc):
        s.relin_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, None, P, [-1, -1], [1, 1], act=f.act)
        s.update_initialization(X0)
    with pytest.raises(capi.AlmpcError):   # nothing solved yet: nothing to advance with
        sw.relin_fnn_advance()
    warm, cold = capi.default_opts(warm_start=1), capi.default_opts()
    x = X0.copy()
    nact = 0
    for step in range(6):
        sw.relin_fnn_step(warm)            # (step 0: no previous step, so it runs cold)
        a = sw.get_results()
        assert np.abs(a["x"][:, :, 0] - x).max() <= 1e-12   # the device's network step = the host's, to rounding
        x = a["x"][:, :, 0].copy()
        sc.update_initialization(x)
        sc.relin_fnn_step(cold)
        b = sc.get_results()
        assert np.all(a["status"] == 0) and np.all(b["status"] == 0), (step, np.bincount(a["status"]), np.bincount(b["status"]))
        # two routes to one optimum.  Compared where the problem means something: a linearisation with spectral radius > 2 sends the
        # predicted states to 1e7 within the horizon, and an input then moves the cost by 1e-17 of its value
        sane = (np.abs(a["e_x"]).reshape(batch, -1).max(axis=1) <= 1e3) & (np.abs(b["e_x"]).reshape(batch, -1).max(axis=1) <= 1e3)
        assert sane.mean() >= 0.98
        assert np.abs(a["u"][sane] - b["u"][sane]).max() <= U_TOL, step
        if step > 0:
            assert np.all(a["iters"] == 0)                  # no ADMM phase ran
            tw, tc = sw.relin_fnn_timing(), sc.relin_fnn_timing()
            # (one inverse instead of two -- which, with a scalar rho, share ONE launch since round 4: the warm design is no longer
            # much shorter, 56 against 60 us at the configs[3] shape; the check is that it is not longer)
            assert tw["design_ms"] < 1.25 * tc["design_ms"]
        for i in range(step, batch, 37):
            if not sane[i]:
                continue
            Ai, Bi = f.jacobian(x[i], u_ref[:, 0])
            p = mo.make_problem(Ai, Bi, N, [-1, -1], [1, 1], x_ref=x_ref, u_ref=u_ref, P=P)
            e = mo.solve_mpc_exact(p, x[i])
            assert np.abs(a["u"][i] - e["u"]).max() <= U_TOL
            nact += ((e["u"] <= -1) | (e["u"] >= 1)).sum()
        # the plant: the network itself, on the device (checked against the host's forward pass at the next step)
        x = np.stack([f.forward(x[i], a["u"][i][:, 0]) for i in range(batch)])
        sw.relin_fnn_advance()
    assert nact > 20
    sw.close(); sc.close()


def test_wave_per_instance_step_equals_the_two_launch_path(capi, mo, monkeypatch):
    """k_step_inst_wave (small per-instance problems: one wave per instance for the whole step) against the k_admm_inst + k_polish
    pair on the same models and states: same certified optimum (the sums of a product run in a different order: 1e-9), same
    statuses; and against the exact oracle on a sample."""
    n, m, N, b = 4, 2, 20, 300
    rng = np.random.default_rng(7)
    f = mo.synthetic_fnn(act="relu")
    x_ref = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N + 1)); u_ref = np.tile(np.array([0.1, -0.2])[:, None], (1, N))
    X0 = x_ref[:, 0][None, :] + mo.splitmix_normal(0x5EED0004, 0, b, n)
    A = np.empty((b, n, n)); B = np.empty((b, n, m))
    for i in range(b):
        A[i], B[i] = f.jacobian(X0[i], u_ref[:, 0])
    Q, R, P = 100.0 * np.eye(n), 0.1 * np.eye(m), 150.0 * np.eye(n)
    res = {}
    for tag, env in (("wave", None), ("pair", "1")):
        if env:
            monkeypatch.setenv("ALMPC_NO_INST_WAVE", env)
        else:
            monkeypatch.delenv("ALMPC_NO_INST_WAVE", raising=False)
        s = capi.Solver(n, m, N, b)
        s.design_batched(A, B, Q, R, None, P, -np.ones(m), np.ones(m))
        s.set_reference(x_ref, u_ref)
        s.update_initialization(X0)
        s.calculate()
        res[tag] = s.get_results()
        # a warm-started second step (the ADMM state of the first) goes through the same kernel
        s.calculate(capi.default_opts(warm_start=1))
        res[tag + "_warm"] = s.get_results()
        s.close()
    monkeypatch.delenv("ALMPC_NO_INST_WAVE", raising=False)
    for a_, b_ in (("wave", "pair"), ("wave_warm", "pair_warm")):
        assert np.array_equal(res[a_]["status"], res[b_]["status"])
        ok = res[a_]["status"] == 0
        assert np.all(ok), np.bincount(res[a_]["status"], minlength=4)
        assert np.abs(res[a_]["u"][ok] - res[b_]["u"][ok]).max() <= 1e-9
        assert np.abs(res[a_]["x"][ok] - res[b_]["x"][ok]).max() <= 1e-8
    assert np.array_equal(res["wave"]["iters"], res["pair"]["iters"])
    for i in range(0, b, 37):
        if res["wave"]["status"][i] == 0:
            pi = mo.make_problem(A[i], B[i], N, -np.ones(m), np.ones(m), x_ref=x_ref, u_ref=u_ref, P=P)
            assert np.abs(res["wave"]["u"][i] - mo.solve_mpc_exact(pi, X0[i])["u"]).max() <= 1e-6


def test_fused_design_chain_equals_the_split_launches(capi, mo, monkeypatch):
    """Per-instance designs with nz <= 64 run the Jacobi scaling as the tail of k_design_instance_t, form V_i inside the first inverse's
    launch and, with a scalar rho, both inverses in one launch (csrc/almpc_api.hip: launch_batched_factor).  The split launches
    (ALMPC_DBG_SPLIT_*) are the same operations on the same operands: same scaling bit for bit, same statuses and iteration counts;
    with the stiffness profile (two inverse launches) as well."""
    n, m, N, b = 4, 2, 20, 200
    f = mo.synthetic_fnn(act="relu")
    x_ref = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N + 1)); u_ref = np.tile(np.array([0.1, -0.2])[:, None], (1, N))
    X0 = x_ref[:, 0][None, :] + mo.splitmix_normal(0x5EED0004, 0, b, n)
    A = np.empty((b, n, n)); B = np.empty((b, n, m))
    for i in range(b):
        A[i], B[i] = f.jacobian(X0[i], u_ref[:, 0])
    Q, R, P = 100.0 * np.eye(n), 0.1 * np.eye(m), 150.0 * np.eye(n)
    switches = ("ALMPC_DBG_SPLIT_SCALE", "ALMPC_DBG_SPLIT_NEGGM", "ALMPC_DBG_SPLIT_INVERSES")
    for profile, rho in (("scalar", 0.1), ("stiffness", 5.0)):
        res = {}
        for tag in ("fused", "split"):
            for k in switches:
                if tag == "split":
                    monkeypatch.setenv(k, "1")
                else:
                    monkeypatch.delenv(k, raising=False)
            s = capi.Solver(n, m, N, b)
            s.design_batched(A, B, Q, R, None, P, -np.ones(m), np.ones(m), rho=rho, rho_profile=profile)
            s.set_reference(x_ref, u_ref)
            s.update_initialization(X0)
            s.calculate(capi.default_opts(rho=rho))
            res[tag] = (s.get_results(), [s.get_design_instance(i) for i in (0, 77, b - 1)])
            s.close()
        for k in switches:
            monkeypatch.delenv(k, raising=False)
        (rf, df), (rs, dsp) = res["fused"], res["split"]
        for a_, b_ in zip(df, dsp):
            assert np.array_equal(a_["d"], b_["d"]) and np.array_equal(a_["H"], b_["H"]) and np.array_equal(a_["F"], b_["F"])
        assert np.array_equal(rf["status"], rs["status"]) and np.all(rf["status"] == 0)
        assert np.array_equal(rf["iters"], rs["iters"]) and np.array_equal(rf["polish_iters"], rs["polish_iters"])
        # V_i sums its products in another order: rounding-level differences, amplified by the conditioning of the instance (the
        # register Gauss-Jordan inverse carries a few eps cond): both forms within 1e-6 of the exact optimum where they differ most
        dev = np.abs(rf["u"] - rs["u"]).reshape(b, -1).max(axis=1)
        assert np.median(dev) <= 1e-11 and dev.max() <= 5e-7
        i = int(np.argmax(dev))
        pi = mo.make_problem(A[i], B[i], N, -np.ones(m), np.ones(m), x_ref=x_ref, u_ref=u_ref, P=P)
        e = mo.solve_mpc_exact(pi, X0[i])["u"]
        assert np.abs(rf["u"][i] - e).max() <= 1e-6 and np.abs(rs["u"][i] - e).max() <= 1e-6
