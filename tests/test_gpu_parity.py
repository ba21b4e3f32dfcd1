"""Parity of the HIP path (through the C ABI) with the oracle.  Run on an MI355X: pytest -m gpu.

Tolerance: BASELINE.json north_star states ||u* - u*_ref||_inf <= 1e-5 (absolute, input units); the tests use
U_TOL = 1e-6 on u (ten times tighter) and X_TOL = 1e-5 on the rolled-out states (the rollout amplifies input errors
through A^k on this marginally unstable plant)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

U_TOL = 1e-6
X_TOL = 1e-5


def make_solver(capi, p, batch, **kw):
    s = capi.Solver(p.n, p.m, p.N, batch, device=0, **kw)
    s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max)
    s.set_reference(p.x_ref, p.u_ref)
    return s


def step(capi, p, X0, opts=None, **kw):
    X0 = np.atleast_2d(np.asarray(X0, dtype=np.float64))
    s = make_solver(capi, p, X0.shape[0], **kw)
    s.update_initialization(X0)
    s.calculate(opts)
    r = s.get_results()
    s.close()
    return r


def _load(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def _problem(mo, g):
    return mo.make_problem(g["A"], g["B"], g["N"], g["u_min"], g["u_max"], x_ref=g["x_ref"], u_ref=g["u_ref"],
                           q=g["q"], r=g["r"], s=g["s"])


# ---------------------------------------------------------------------------- design kernels
@pytest.mark.parametrize("which", ["di", "di_S", "qtp", "quad"])
def test_design_matches_oracle(capi, mo, qtp_ab, which):
    p = {"di": mo.double_integrator, "quad": mo.quadrotor,
         "di_S": lambda: _problem(mo, _load("double_integrator_S")),
         "qtp": lambda: mo.qtp_linear_fixture_problem(*qtp_ab)}[which]()
    s = make_solver(capi, p, 1)
    d = s.get_design()
    s.close()
    od = mo.design_shared(p)
    assert np.abs(d["P"] - p.P).max() <= 1e-9 * np.abs(p.P).max()           # DARE (src/sub/design_mpc.jl:327)
    assert np.abs(d["H"] - od["H"]).max() <= 1e-10 * np.abs(od["H"]).max()  # H = 2(Gamma'Qbar Gamma + Rbar + D'SbarD)
    assert np.abs(d["F"] - od["F"]).max() <= 1e-10 * np.abs(od["F"]).max()  # F = 2 Gamma'Qbar Phi
    np.testing.assert_allclose(d["d"], od["d"], rtol=1e-10)
    assert np.allclose(d["H"], d["H"].T, rtol=0, atol=1e-9 * np.abs(d["H"]).max())


def test_user_supplied_terminal_weight(capi, mo):
    p = mo.double_integrator()
    P = np.array([[3.0, 0.5], [0.5, 2.0]])
    p2 = mo.make_problem(p.A, p.B, p.N, p.u_min, p.u_max, P=P)
    s = capi.Solver(2, 1, 10, 1)
    s.design_shared(p.A, p.B, p.Q, p.R, None, P, p.u_min, p.u_max)
    d = s.get_design()
    s.update_initialization([[4.0, 1.0]])
    s.calculate()
    r = s.get_results()
    s.close()
    assert np.abs(d["H"] - mo.design_shared(p2)["H"]).max() <= 1e-10 * np.abs(d["H"]).max()
    assert np.abs(r["u"][0] - mo.solve_mpc_exact(p2, np.array([4.0, 1.0]))["u"]).max() <= U_TOL


# ---------------------------------------------------------------------------- golden vectors through the C ABI
@pytest.mark.parametrize("name", ["double_integrator", "double_integrator_S", "qtp_linear", "quadrotor"])
def test_golden_vectors(capi, mo, name):
    g = _load(name)
    p = _problem(mo, g)
    X0 = np.array([c["x0"] for c in g["cases"]])
    r = step(capi, p, X0)
    assert np.all(r["status"] == 0)
    for i, c in enumerate(g["cases"]):
        assert np.abs(r["u"][i] - np.array(c["u"])).max() <= U_TOL
        assert np.abs(r["x"][i] - np.array(c["x"])).max() <= X_TOL
    np.testing.assert_allclose(r["e_u"], r["u"] - p.u_ref[None], atol=1e-14)
    np.testing.assert_allclose(r["e_x"], r["x"] - p.x_ref[None], atol=1e-12)
    np.testing.assert_array_equal(r["x"][:, :, 0], X0)  # x[:,1] is fixed to x0


def test_reference_scenario_assertions(capi, mo, qtp_ab):
    """test/computation_mpc_test.jl:1053-1054 on the reference's own fixture."""
    r = step(capi, mo.qtp_linear_fixture_problem(*qtp_ab), np.full((1, 4), 0.6))
    assert np.all(np.abs(r["x"][0] - 0.65) <= 0.5) and np.all(np.abs(r["u"][0][:, 0] - 1.2) <= 3.0)
    assert np.all(r["x"] != 0) and np.all(r["u"] != 0)  # the `!= 0` assertions of test/terminal_ingredient_test.jl:162-168


# ---------------------------------------------------------------------------- seeded batches vs the oracle
@pytest.mark.parametrize("amp", [0.3, 1.0, 3.0])
@pytest.mark.parametrize("batch", [1, 15, 16, 17, 100])
def test_quadrotor_batches_vs_exact_oracle(capi, mo, amp, batch):
    p = mo.quadrotor()
    X0 = mo.quadrotor_x0_batch(batch, amp, first_instance=1000)
    r = step(capi, p, X0)
    assert np.all(r["status"] == 0)
    for i in range(batch):
        e = mo.solve_mpc_exact(p, X0[i])
        assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL
        assert np.abs(r["x"][i] - e["x"]).max() <= X_TOL
    assert np.all(r["u"] <= p.u_max[None, :, None] + 1e-15) and np.all(r["u"] >= p.u_min[None, :, None] - 1e-15)


def test_algorithm_parity_with_c_restatement(capi, mo, co):
    """Same algorithm, same options: ADMM iteration counts, polish sweep counts and iterates agree with the C
    restatement (the HIP kernel tracks H'x by the KKT identity, the oracle multiplies: counts may differ only if a
    residual sits within rounding of its threshold -- allowed on <= 0.2 % of the instances)."""
    p = mo.quadrotor()
    X0 = np.concatenate([mo.quadrotor_x0_batch(512, a, first_instance=512 * k) for k, a in enumerate((0.3, 1.0, 3.0))])
    des = mo.design_shared(p)
    for max_iter in (25, 50, 200):
        r = step(capi, p, X0, capi.default_opts(max_iter=max_iter))
        c = co.step_batch(p, des, X0, max_iter=max_iter)
        assert (r["iters"] != c["iters"]).mean() <= 2e-3
        assert (r["polish_iters"] != c["polish_iters"]).mean() <= 2e-3
        assert np.array_equal(r["status"], c["status"])
        assert np.abs(r["u"] - c["u"]).max() <= U_TOL and np.abs(r["x"] - c["x"]).max() <= X_TOL


def test_stiffness_rho_profile_parity(capi, mo, co):
    """almpc_set_rho_profile(1): rho_i = rho / (H'^-1)_ii.  Same algorithm parity with the C restatement (ADMM iterates with
    polish off; iteration counts and results with polish on) and the exact optimum."""
    p = mo.quadrotor()
    X0 = np.concatenate([mo.quadrotor_x0_batch(160, a, first_instance=160 * k) for k, a in enumerate((0.3, 1.0, 3.0, 4.0))])
    des = mo.design_shared(p, rho=30.0, rho_profile="stiffness")
    s = capi.Solver(p.n, p.m, p.N, len(X0))
    s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, rho=30.0, rho_profile="stiffness")
    s.set_reference(p.x_ref, p.u_ref)
    s.update_initialization(X0)
    for kw in (dict(max_iter=8, check_every=8, polish=0), dict(max_iter=8, check_every=8), dict(max_iter=25)):
        s.calculate(capi.default_opts(rho=30.0, **kw))
        r = s.get_results()
        c = co.step_batch(p, des, X0, max_iter=kw["max_iter"], check_every=kw.get("check_every", 25), polish=bool(kw.get("polish", 1)))
        assert np.array_equal(r["iters"], c["iters"])
        if kw.get("polish", 1):
            assert np.all(r["status"] == 0) and (r["polish_iters"] != c["polish_iters"]).mean() <= 5e-3
        assert np.abs(r["u"] - c["u"]).max() <= (1e-9 if not kw.get("polish", 1) else U_TOL)
    s.close()
    for i in range(0, len(X0), 16):
        assert np.abs(r["u"][i] - mo.solve_mpc_exact(p, X0[i])["u"]).max() <= U_TOL


def test_admm_only_matches_oracle_iterate(capi, mo, co):
    """polish off: the result is the ADMM iterate z itself -> compare the iterates of the two implementations."""
    p = mo.quadrotor()
    X0 = mo.quadrotor_x0_batch(64, 3.0)
    des = mo.design_shared(p)
    for kw in (dict(max_iter=25), dict(max_iter=100, check_every=10), dict(max_iter=37, check_every=5, alpha=1.0)):
        r = step(capi, p, X0, capi.default_opts(polish=0, **kw))
        c = co.step_batch(p, des, X0, polish=False, **kw)
        assert np.array_equal(r["iters"], c["iters"]) and np.array_equal(r["status"], c["status"])
        assert np.abs(r["u"] - c["u"]).max() <= 1e-9
        assert np.all(r["polish_iters"] == 0)


def test_osqp_default_settings_converge(capi, mo):
    """OSQP's own defaults (max_iter 4000): every instance meets the ADMM tolerance on the double integrator."""
    p = mo.double_integrator()
    X0 = np.array([[1.0, 0.0], [5.0, 0.0], [-8.0, 2.0]])
    r = step(capi, p, X0, capi.default_opts(max_iter=4000, polish=0))
    assert np.all(r["status"] == 0) and np.all(r["iters"] % 25 == 0) and np.all(r["iters"] < 4000)
    for i in range(3):
        assert np.abs(r["u"][i] - mo.solve_mpc_exact(p, X0[i])["u"]).max() <= 0.5  # eps = 1e-3 quality only (cf. the reference's atol = 3)


# ---------------------------------------------------------------------------- odd shapes
@pytest.mark.parametrize("n,m,N", [(1, 1, 1), (1, 1, 7), (3, 2, 7), (5, 3, 9), (2, 1, 128), (7, 5, 25), (16, 8, 16), (20, 1, 40), (64, 2, 3),
                                   (3, 1, 117), (2, 1, 113), (6, 3, 39), (4, 4, 31)])  # the last four: 8-wave tiles -> fused step kernel, odd nz
def test_random_stable_plants_of_many_shapes(capi, mo, n, m, N):
    """nz = m*N from 1 to 128 (odd values too), n up to 64 (n = 20, 64: the separate rollout kernels): random plants
    with spectral radius 0.97, bounds tight enough to be active, batch not a multiple of anything."""
    rng = np.random.default_rng(1000 * n + 10 * m + N)
    A = rng.standard_normal((n, n))
    A *= 0.97 / np.max(np.abs(np.linalg.eigvals(A)))
    B = rng.standard_normal((n, m))
    p = mo.make_problem(A, B, N, -0.5 * np.ones(m), 0.7 * np.ones(m), x_ref=0.1 * rng.standard_normal(n), u_ref=0.05 * rng.standard_normal(m),
                        q=10.0, r=1.0, s=0.5 if N > 2 else 0.0)
    batch = 37
    X0 = 3.0 * rng.standard_normal((batch, n))
    r = step(capi, p, X0)
    assert np.all(r["status"] == 0)
    nact = 0
    for i in range(0, batch, 3):
        e = mo.solve_mpc_exact(p, X0[i])
        assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL and np.abs(r["x"][i] - e["x"]).max() <= 10 * X_TOL
        nact += (np.isclose(e["u"], p.u_min[:, None] + p.u_ref * 0) | np.isclose(e["u"], p.u_max[:, None])).sum()
    assert nact > 0


# ---------------------------------------------------------------------------- edge cases
def test_heavy_saturation_uses_second_polish_tier(capi, mo):
    """Amplitude 10: working sets beyond 32 rows leave the LDS mode of the polish and continue in its global mode."""
    p = mo.quadrotor()
    X0 = mo.quadrotor_x0_batch(48, 10.0, first_instance=300)
    r = step(capi, p, X0)
    nact = [(np.isclose(r["u"][i], p.u_min[:, None]) | np.isclose(r["u"][i], p.u_max[:, None])).sum() for i in range(48)]
    assert max(nact) > 32, "test input does not force the overflow branch"
    solved = r["status"] == 0
    assert solved.sum() >= 46
    for i in range(48):
        e = mo.solve_mpc_exact(p, X0[i])
        if solved[i]:
            assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL
        else:  # only a working set that outgrows the 64-row capacity may be left to the ADMM iterate
            na = (np.isclose(e["u"], p.u_min[:, None]) | np.isclose(e["u"], p.u_max[:, None])).sum()
            assert na >= 60 and r["status"][i] == 1


def test_working_set_beyond_largest_tier_keeps_admm_iterate(capi, mo):
    """More than 64 active bounds: no polish tier takes the instance; it keeps the (feasible) ADMM iterate and is
    reported as ALMPC_MAX_ITER unless ADMM itself met its tolerance."""
    p = mo.quadrotor()
    X0 = mo.quadrotor_x0_batch(8, 100.0)
    r = step(capi, p, X0, capi.default_opts(max_iter=25))
    assert np.all(r["u"] <= p.u_max[None, :, None]) and np.all(r["u"] >= p.u_min[None, :, None])
    assert np.all(np.isin(r["status"], (0, 1)))


def test_all_inputs_inside_bounds_and_zero_state(capi, mo):
    p = mo.quadrotor()
    r = step(capi, p, np.zeros((3, 12)))
    assert np.all(r["status"] == 0) and np.abs(r["u"]).max() <= 1e-12 and np.abs(r["x"]).max() <= 1e-12


def test_non_finite_input_is_flagged(capi, mo):
    p = mo.quadrotor()
    X0 = mo.quadrotor_x0_batch(20, 1.0)
    X0[3, 2] = np.nan
    X0[18, 0] = np.inf
    r = step(capi, p, X0)
    assert r["status"][3] == 2 and r["status"][18] == 2
    ok = np.ones(20, bool)
    ok[[3, 18]] = False
    assert np.all(r["status"][ok] == 0)
    for i in np.flatnonzero(ok)[:6]:
        assert np.abs(r["u"][i] - mo.solve_mpc_exact(p, X0[i])["u"]).max() <= U_TOL


def test_nonzero_and_per_instance_references(capi, mo, qtp_ab):
    A, B = qtp_ab
    p = mo.qtp_linear_fixture_problem(A, B)
    batch = 5
    rng = np.random.default_rng(3)
    X0 = 0.6 + 0.1 * rng.standard_normal((batch, 4))
    xr = 0.65 + 0.05 * rng.standard_normal((batch, 4, 1)) * np.ones((1, 1, 6))
    ur = 1.2 + 0.5 * rng.standard_normal((batch, 2, 1)) * np.ones((1, 1, 5))
    s = make_solver(capi, p, batch)
    s.set_reference(xr, ur, per_instance=True)
    s.update_initialization(X0)
    s.calculate()
    r = s.get_results()
    s.close()
    for i in range(batch):
        pi = mo.make_problem(A, B, 5, p.u_min, p.u_max, x_ref=xr[i], u_ref=ur[i])
        e = mo.solve_mpc_exact(pi, X0[i])
        assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL and np.abs(r["x"][i] - e["x"]).max() <= X_TOL
        np.testing.assert_allclose(r["e_u"][i], r["u"][i] - ur[i], atol=1e-14)


def test_input_rate_weight_with_time_varying_reference(capi, mo):
    """S != 0 penalises u[:,i]-u[:,i+1] on u (not e_u): with a time-varying u_ref it adds a linear term."""
    ur = np.linspace(-0.3, 0.4, 8).reshape(1, 8)
    p = mo.make_problem([[1.0, 1.0], [0.0, 1.0]], [[0.5], [1.0]], 8, [-1.0], [1.0], x_ref=np.zeros((2, 9)), u_ref=ur, s=2.0)
    X0 = np.array([[2.0, 0.5], [-1.0, 0.2]])
    r = step(capi, p, X0)
    for i in range(2):
        assert np.abs(r["u"][i] - mo.solve_mpc_exact(p, X0[i])["u"]).max() <= U_TOL


def test_input_rate_weight_is_dropped_together_with_R(capi, mo):
    """The reference's branch rule (src/sub/design_mpc.jl:423-466): with R[1,1] == 0 the objective has neither the input term nor the
    input-rate term, whatever S is -- also not the linear term a time-varying u_ref would otherwise add (R = 0, S != 0, u_ref varying).
    P is given: the DARE needs R > 0."""
    ur = np.linspace(-0.3, 0.4, 8).reshape(1, 8)
    P = np.array([[300.0, 40.0], [40.0, 150.0]])
    A, B = [[0.9, 1.0], [0.0, 0.8]], [[0.5], [1.0]]
    p = mo.make_problem(A, B, 8, [-1.0], [1.0], x_ref=np.zeros((2, 9)), u_ref=ur, r=0.0, s=2.0, P=P)
    p_noS = mo.make_problem(A, B, 8, [-1.0], [1.0], x_ref=np.zeros((2, 9)), u_ref=ur, r=0.0, s=0.0, P=P)
    X0 = np.array([[2.0, 0.5], [-1.0, 0.2], [0.3, -0.1]])
    s = capi.Solver(2, 1, 8, len(X0))
    s.design_shared(p.A, p.B, p.Q, p.R, p.S, P, p.u_min, p.u_max)
    s.set_reference(p.x_ref, p.u_ref)
    s.update_initialization(X0)
    s.calculate()
    r = s.get_results()
    d = s.get_design()
    s.close()
    od = mo.design_shared(p_noS)
    assert np.abs(d["H"] - od["H"]).max() <= 1e-10 * np.abs(od["H"]).max()
    assert np.all(r["status"] == 0)
    for i in range(len(X0)):
        for q in (p, p_noS):  # the oracle applies the same rule: both statements have the same optimum
            assert np.abs(r["u"][i] - mo.solve_mpc_exact(q, X0[i])["u"]).max() <= U_TOL


def test_warm_start_closed_loop(capi, mo):
    """Receding horizon: 30 closed-loop steps with warm-started ADMM reach the reference and agree with the oracle."""
    p = mo.quadrotor()
    batch = 32
    X = mo.quadrotor_x0_batch(batch, 1.0, first_instance=77)
    s = make_solver(capi, p, batch)
    warm = capi.default_opts(warm_start=1)
    for k in range(30):
        s.update_initialization(X)
        s.calculate(warm if k else None)
        r = s.get_results(want=("u", "status"))
        assert np.all(r["status"] == 0)
        if k % 10 == 0:
            for i in range(0, batch, 8):
                assert np.abs(r["u"][i] - mo.solve_mpc_exact(p, X[i])["u"]).max() <= U_TOL
        X = X @ p.A.T + r["u"][:, :, 0] @ p.B.T
    s.close()
    assert np.abs(X[:, :3]).max() < 0.5 * np.abs(mo.quadrotor_x0_batch(batch, 1.0, first_instance=77)[:, :3]).max()


def test_closed_loop_on_device(capi, mo):
    """almpc_advance_plant: 25 receding-horizon steps entirely on the device (warm-started ADMM) equal the host loop
    x+ = A x + B u[:,1] driven by the exact oracle at every step (within tolerance accumulation)."""
    p = mo.quadrotor()
    batch = 24
    X = mo.quadrotor_x0_batch(batch, 2.0, first_instance=5)
    s = make_solver(capi, p, batch)
    s.update_initialization(X)
    Xh = X.copy()
    for k in range(25):
        s.calculate(capi.default_opts(warm_start=1) if k else None, sync=False)
        s.advance_plant()
        if k in (0, 7, 24):
            r = s.get_results(want=("u", "x", "status"))
            assert np.all(r["status"] == 0)
            np.testing.assert_allclose(r["x"][:, :, 0], Xh, atol=1e-8)          # the device x0 is the host-propagated state
            for i in range(0, batch, 6):
                assert np.abs(r["u"][i] - mo.solve_mpc_exact(p, Xh[i])["u"]).max() <= U_TOL
        u0 = np.array([mo.solve_mpc_exact(p, Xh[i])["u"][:, 0] for i in range(batch)])
        Xh = Xh @ p.A.T + u0 @ p.B.T
    s.synchronize()
    s.close()
    assert np.abs(Xh[:, :3]).max() < np.abs(X[:, :3]).max()


def test_api_error_behaviour(capi, mo):
    p = mo.double_integrator()
    s = capi.Solver(2, 1, 10, 2)
    with pytest.raises(capi.AlmpcError) as ei:
        s.calculate()
    assert ei.value.code == -5  # not designed
    with pytest.raises(capi.AlmpcError) as ei:
        s.design_shared(p.A, p.B, p.Q, p.R, None, None, p.u_min, p.u_max, xmin=[-1, -1])
    assert ei.value.code == -1  # xmin without xmax
    with pytest.raises(capi.AlmpcError) as ei:
        s.design_shared(p.A, p.B, p.Q, p.R, None, None, p.u_min, p.u_max, xmin=[1, 1], xmax=[-1, -1])
    assert ei.value.code == -1
    with pytest.raises(capi.AlmpcError) as ei:
        s.design_shared(p.A, p.B, p.Q, p.R, None, None, [1.0], [-1.0])
    assert ei.value.code == -1
    with pytest.raises(capi.AlmpcError) as ei:
        s.design_shared(p.A, p.B, p.Q, 0.0 * p.R, None, None, p.u_min, p.u_max, rho=-1.0)
    assert ei.value.code == -1
    s.design_shared(p.A, p.B, p.Q, p.R, None, None, p.u_min, p.u_max)
    with pytest.raises(capi.AlmpcError) as ei:
        s.calculate(capi.default_opts(rho=0.5))  # differs from the design rho
    assert ei.value.code == -1
    with pytest.raises(capi.AlmpcError):
        s.calculate(capi.default_opts(alpha=2.5))
    s.update_initialization([[1.0, 0.0], [5.0, 0.0]])
    s.calculate()
    assert np.all(s.get_results()["status"] == 0)
    s.close()


# ---------------------------------------------------------------------------- state rows: state box, terminal equality
def _constrained_problems(mo):
    A, B = [[1.0, 1.0], [0.0, 1.0]], [[0.5], [1.0]]
    q = mo.quadrotor()
    xm = np.array([50, 50, 50, 1.0, 1.0, 1.0, 0.3, 0.3, 0.3, 2, 2, 2.0])
    return {
        "di_box": (mo.make_problem(A, B, 10, [-1.0], [1.0], x_min=[-10.0, -0.8], x_max=[10.0, 0.8]),
                   np.array([[5.0, 0.0], [-6.0, 0.5], [1.0, 0.0], [0.0, 0.0], [20.0, 0.0], [3.0, 0.7], [0.0, 0.9]])),
        "di_eq": (mo.make_problem(A, B, 10, [-1.0], [1.0], terminal="equality"),
                  np.array([[2.0, 0.0], [1.0, -0.5], [30.0, 0.0], [0.0, 0.0]])),
        "di_box_eq": (mo.make_problem(A, B, 10, [-1.0], [1.0], x_min=[-10.0, -0.8], x_max=[10.0, 0.8], terminal="equality"),
                      np.array([[2.0, 0.0], [1.0, -0.5], [3.0, 0.5], [0.0, 0.0]])),
        "quad_box": (mo.make_problem(q.A, q.B, 30, q.u_min, q.u_max, x_min=-xm, x_max=xm),
                     np.concatenate([mo.quadrotor_x0_batch(40, 1.0), mo.quadrotor_x0_batch(24, 0.5, first_instance=500)])),
        "quad_eq": (mo.make_problem(q.A, q.B, 30, q.u_min, q.u_max, terminal="equality"), mo.quadrotor_x0_batch(24, 0.1)),
    }


@pytest.mark.parametrize("name", ["di_box", "di_eq", "di_box_eq", "quad_box", "quad_eq"])
def test_state_rows_vs_exact_oracle(capi, mo, name):
    """State box (kw mpc_state_constraint, ..linear.jl:62-70) and terminal equality (src/sub/design_mpc.jl:330-331):
    solved instances match the KKT-certified oracle; infeasible ones (x0 outside the box, unreachable terminal set)
    are reported as ALMPC_INFEASIBLE exactly where the oracle finds no feasible point."""
    p, X0 = _constrained_problems(mo)[name]
    s = capi.Solver(p.n, p.m, p.N, len(X0))
    s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=p.x_min, xmax=p.x_max, terminal=p.terminal)
    s.set_reference(p.x_ref, p.u_ref)
    s.update_initialization(X0)
    s.debug_poison_lds()
    s.calculate()
    r = s.get_results()
    s.close()
    n_state_active = 0
    for i in range(len(X0)):
        try:
            e = mo.solve_mpc_exact(p, X0[i], return_info=True)
        except ValueError:
            assert r["status"][i] == 3
            continue
        assert r["status"][i] == 0
        assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL and np.abs(r["x"][i] - e["x"]).max() <= X_TOL
        n_state_active += e["info"]["n_active_state"]
        if p.x_min is not None:
            assert np.all(r["x"][i] <= p.x_max[:, None] + 1e-9) and np.all(r["x"][i] >= p.x_min[:, None] - 1e-9)
        if p.terminal == "equality":
            assert np.abs(r["e_x"][i][:, -1]).max() <= 1e-9
    assert n_state_active > 0, "test inputs never activate a state row"


def test_state_rows_working_sets_beyond_32_rows(capi, mo):
    """Tight state box + terminal equality on the quadrotor: 12 equality rows are always in the working set, and several instances
    end with 33..64 active rows.  Those outgrow the 32-row build of k_polish_gen, are flagged and redone by its 64-row build in a
    second launch: every feasible instance must match the exact oracle, every infeasible one must be reported as such."""
    q = mo.quadrotor()
    xmax = np.array([1, 1, 1, .5, .5, .5, .1, .1, .1, .1, .1, .1]) * 3.0
    p = mo.make_problem(q.A, q.B, 30, q.u_min, q.u_max, x_min=-xmax, x_max=xmax, terminal="equality")
    X0 = np.clip(mo.quadrotor_x0_batch(32, 1.0, first_instance=900), -0.99 * xmax, 0.99 * xmax)
    s = capi.Solver(p.n, p.m, p.N, len(X0))
    s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=p.x_min, xmax=p.x_max, terminal="equality")
    s.update_initialization(X0)
    s.calculate()
    r = s.get_results()
    s.close()
    big = 0
    for i in range(len(X0)):
        try:
            e = mo.solve_mpc_exact(p, X0[i], return_info=True)
        except ValueError:
            assert r["status"][i] == 3
            continue
        assert r["status"][i] == 0, (i, r["status"][i])
        assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL and np.abs(r["x"][i] - e["x"]).max() <= X_TOL
        assert np.abs(r["e_x"][i][:, -1]).max() <= 1e-9
        na = int((np.isclose(e["u"], p.u_min[:, None]) | np.isclose(e["u"], p.u_max[:, None])).sum()) + e["info"]["n_active_state"]
        big += na > 32
    assert big >= 3, "test inputs do not reach the second tier"


def test_state_rows_need_polish(capi, mo):
    p, X0 = _constrained_problems(mo)["di_box"]
    s = capi.Solver(2, 1, 10, len(X0))
    s.design_shared(p.A, p.B, p.Q, p.R, None, None, p.u_min, p.u_max, xmin=p.x_min, xmax=p.x_max)
    s.update_initialization(X0)
    with pytest.raises(capi.AlmpcError) as ei:
        s.calculate(capi.default_opts(polish=0))
    assert ei.value.code == -4
    s.close()


def test_state_constraint_mirror(pkg, mo):
    """Host mirror: the state box exists only if kw mpc_state_constraint is PRESENT; terminal "equality"; an
    infeasible instance raises like JuMP.value would."""
    p, X0 = _constrained_problems(mo)["di_box"]
    sys_ = pkg.ConstrainedLinearControlDiscreteSystem(p.A, p.B, pkg.Hyperrectangle(p.x_min, p.x_max), pkg.Hyperrectangle(p.u_min, p.u_max))
    C = pkg.proceed_controller(sys_, "model_predictive_control", 10, 1, [0.0, 0.0], [0.0], mpc_state_constraint=True)
    pkg.update_initialization(C, [5.0, 0.0])
    pkg.calculate(C)
    e = mo.solve_mpc_exact(p, np.array([5.0, 0.0]))
    assert np.abs(C.computation_results.u - e["u"]).max() <= U_TOL
    assert C.computation_results.x[1].min() >= -0.8 - 1e-9
    pkg.update_initialization(C, [20.0, 0.0])  # outside the state box: stage 1 is constrained too
    with pytest.raises(ArithmeticError):
        pkg.calculate(C)
    C.tuning.modeler.solver.close()
    # without the kw the same system has no state rows (the reference reads only the presence of the key)
    C2 = pkg.proceed_controller(sys_, "model_predictive_control", 10, 1, [0.0, 0.0], [0.0])
    pkg.update_initialization(C2, [5.0, 0.0])
    pkg.calculate(C2)
    assert C2.computation_results.x[1].min() < -0.8
    C2.tuning.modeler.solver.close()
    C3 = pkg.proceed_controller(sys_, "model_predictive_control", 10, 1, [0.0, 0.0], [0.0], mpc_terminal_ingredient="equality")
    assert C3.tuning.terminal_ingredient.Xf == "equality"
    pkg.update_initialization(C3, [2.0, 0.0])
    pkg.calculate(C3)
    assert np.abs(C3.computation_results.e_x[:, -1]).max() <= 1e-9
    C3.tuning.modeler.solver.close()


# ---------------------------------------------------------------------------- black-box (Fnn) models: BASELINE config 4
@pytest.mark.parametrize("build", ["wave_per_point", "workgroup_per_point"])
def test_fnn_jacobian_kernel_vs_oracle(capi, mo, build, monkeypatch):
    """Batched linearisation on the GPU (stand-in for AutomationLabsSystems.proceed_system_linearization) vs the numpy
    restatement, at random points on both sides of the relu kinks, for every supported activation (device tanh/exp vs
    libm: a few ulp through two hidden layers); both kernel builds (small networks: one wave per point, weights in LDS)."""
    if build == "workgroup_per_point":
        monkeypatch.setenv("ALMPC_FNN_WG", "1")
    for act in ("relu", "identity", "tanh", "sigmoid", "swish"):
        f = mo.synthetic_fnn(act=act)
        X = mo.splitmix_normal(0x5EED0004, 0, 200, 4) * 2.0
        U = mo.splitmix_normal(0x5EED0005, 0, 200, 2)
        A, B, fx = capi.fnn_linearize(f.W_in, f.W_h, f.b_h, f.W_out, X, U, act=act, want_f=True)
        for i in range(0, 200, 3):
            Ao, Bo = f.jacobian(X[i], U[i])
            assert np.abs(A[i] - Ao).max() <= 1e-13 and np.abs(B[i] - Bo).max() <= 1e-13, act
            assert np.abs(fx[i] - f.forward(X[i], U[i])).max() <= 1e-13, act
    # deeper / wider network, no hidden layer edge case
    f = mo.synthetic_fnn(n=3, m=1, H=40, L=4)
    A, B = capi.fnn_linearize(f.W_in, f.W_h, f.b_h, f.W_out, np.ones((2, 3)) * 0.3, np.ones((2, 1)) * -0.2)
    Ao, Bo = f.jacobian(np.ones(3) * 0.3, np.ones(1) * -0.2)
    assert np.abs(A[1] - Ao).max() <= 1e-13 and np.abs(B[0] - Bo).max() <= 1e-13
    f0 = mo.FnnModel(f.W_in, [], [], f.W_out)
    A, B = capi.fnn_linearize(f0.W_in, [], [], f0.W_out, np.zeros((1, 3)), np.zeros((1, 1)))
    assert np.abs(A[0] - (f0.W_out @ f0.W_in)[:, :3]).max() <= 1e-12 * np.abs(A[0]).max()


def test_config4_fnn_linearised_mpc(pkg, capi, mo):
    """BASELINE configs[3] in the reference's own semantics: Fnn model, LinearProgramming branch = linearise at the first
    reference, P from the linearisation at the last reference, then the condensed QP; N = 20, batch = 1024."""
    f = mo.synthetic_fnn()
    sys_ = pkg.ConstrainedBlackBoxControlDiscreteSystem(pkg.Fnn(f.W_in, f.W_h, f.b_h, f.W_out, f.act), 4, 2,
                                                        pkg.Hyperrectangle([-10] * 4, [10] * 4), pkg.Hyperrectangle([-1, -1], [1, 1]))
    x_ref, u_ref = [0.2, -0.1, 0.05, 0.0], [0.1, -0.2]
    batch = 1024
    C = pkg.proceed_controller(sys_, "model_predictive_control", 20, 1, x_ref, u_ref, mpc_batch=batch)
    p = mo.fnn_linear_problem(f, 20, [-1, -1], [1, 1], x_ref, u_ref)
    assert np.abs(C.tuning.terminal_ingredient.P - p.P).max() <= 1e-9 * np.abs(p.P).max()
    X0 = np.asarray(x_ref)[None, :] + mo.splitmix_normal(0x5EED0004, 0, batch, 4) * 2.0
    res = pkg._model_predictive_control_computation(C, X0)
    assert res.u.shape == (batch, 2, 20) and res.x.shape == (batch, 4, 21)
    nact = 0
    for i in range(0, batch, 16):
        e = mo.solve_mpc_exact(p, X0[i])
        assert np.abs(res.u[i] - e["u"]).max() <= U_TOL and np.abs(res.x[i] - e["x"]).max() <= X_TOL
        nact += ((e["u"] <= -1) | (e["u"] >= 1)).sum()
    assert nact > 50
    C.tuning.modeler.solver.close()


def test_dare_export(capi, mo, qtp_ab):
    import scipy.linalg as sla
    A, B = qtp_ab
    P = capi.dare(A, B, 100 * np.eye(4), 0.1 * np.eye(2))
    assert np.abs(P - sla.solve_discrete_are(A, B, 100 * np.eye(4), 0.1 * np.eye(2))).max() <= 1e-9 * np.abs(P).max()


# ---------------------------------------------------------------------------- host mirror of the reference API
def test_proceed_controller_mirror(pkg, mo, qtp_ab):
    """proceed_controller -> update_initialization! -> calculate! with the reference's argument order and result
    shapes (test/computation_mpc_test.jl:981-1054; design asserts of test/design_mpc_implementation_test.jl:82-115)."""
    A, B = qtp_ab
    sys_ = pkg.ConstrainedLinearControlDiscreteSystem(A, B, pkg.Hyperrectangle([0.2] * 4, [1.36, 1.36, 1.30, 1.30]),
                                                      pkg.Hyperrectangle([0, 0], [4, 3.26]))
    C = pkg.proceed_controller(sys_, "model_predictive_control", 5, 5, [0.65] * 4, [1.2] * 2, mpc_terminal_ingredient="none")
    assert C.tuning.horizon == 5 and C.tuning.sample_time == 5.0 and C.tuning.max_time == 30
    assert C.tuning.terminal_ingredient.Xf == "none" and np.all(C.tuning.terminal_ingredient.P != 0)
    assert C.computation_results.x.shape == (4, 6) and C.computation_results.u.shape == (2, 5)
    pkg.update_initialization(C, [0.6] * 4)
    pkg.calculate(C)
    e = mo.solve_mpc_exact(mo.qtp_linear_fixture_problem(A, B), np.full(4, 0.6))
    res = C.computation_results
    assert np.abs(res.u - e["u"]).max() <= U_TOL and np.abs(res.x - e["x"]).max() <= X_TOL
    assert np.abs(res.e_x - e["e_x"]).max() <= X_TOL and np.abs(res.e_u - e["e_u"]).max() <= U_TOL
    assert np.all(np.abs(res.x - 0.65) <= 0.5) and np.all(np.abs(res.u[:, 0] - 1.2) <= 3)
    C.tuning.modeler.solver.close()


def test_batched_computation_mirror(pkg, mo):
    p = mo.quadrotor()
    sys_ = pkg.ConstrainedLinearControlDiscreteSystem(p.A, p.B, pkg.Hyperrectangle([-100] * 12, [100] * 12),
                                                      pkg.Hyperrectangle(p.u_min, p.u_max))
    C = pkg.proceed_controller(sys_, "model_predictive_control", 30, 1, [0.0] * 12, [0.0] * 4, mpc_batch=40, mpc_solver="hip")
    X0 = mo.quadrotor_x0_batch(40, 3.0)
    res = pkg._model_predictive_control_computation(C, X0)
    assert res.u.shape == (40, 4, 30) and res.x.shape == (40, 12, 31)
    for i in range(0, 40, 5):
        assert np.abs(res.u[i] - mo.solve_mpc_exact(p, X0[i])["u"]).max() <= U_TOL
    C.tuning.modeler.solver.close()


# ---------------------------------------------------------------------------- full size: size-independent properties
def test_full_size_batch_properties(capi, mo):
    """BASELINE configs[1] size (4096 instances): KKT certificate of every instance (scaled coordinates), dynamics
    consistency of the rollout, and agreement with the exact oracle on a sample."""
    p = mo.quadrotor()
    batch = 4096
    amp = np.array([0.3, 1.0, 3.0])[np.arange(batch) % 3]
    X0 = mo.splitmix_normal(0x5EED0002, 0, batch, 12) * mo.QUADROTOR_X0_SCALE[None] * amp[:, None]
    r = step(capi, p, X0)
    assert np.all(r["status"] == 0)
    des = mo.design_shared(p)
    d = des["d"]
    W = (r["e_u"].transpose(0, 2, 1).reshape(batch, -1)) / d[None]
    Fs = X0 @ des["Fs"].T
    Gd = W @ des["Hs"] + Fs                                    # gradient H'w + f' per instance
    kkt = np.abs(W - np.clip(W - Gd, des["lo"][None], des["hi"][None])).max(axis=1)
    assert kkt.max() <= 1e-8 * max(1.0, np.abs(Fs).max())
    # rollout obeys the deviation dynamics e_x[:,k+1] = A e_x[:,k] + B e_u[:,k]  (..linear.jl:58-60)
    ex, eu = r["e_x"], r["e_u"]
    pred = np.einsum("ij,bjk->bik", p.A, ex[:, :, :-1]) + np.einsum("ij,bjk->bik", p.B, eu)
    assert np.abs(pred - ex[:, :, 1:]).max() <= 1e-9 * max(1.0, np.abs(ex).max())
    for i in range(0, batch, 128):
        assert np.abs(r["u"][i] - mo.solve_mpc_exact(p, X0[i])["u"]).max() <= U_TOL


@pytest.mark.parametrize("max_iter", [1, 3, 25])
def test_poisoned_lds_and_empty_initial_guess(capi, mo, max_iter):
    """Every kernel must initialise the LDS it reads: the LDS of all CUs is filled with NaN patterns before each solve.
    max_iter = 1 also forces the polish to start from an (almost) empty working set and add every active row itself
    (the branch that once read uninitialised Sinv padding)."""
    p = mo.quadrotor()
    X0 = np.concatenate([mo.quadrotor_x0_batch(64, a, first_instance=40 * k) for k, a in enumerate((0.3, 1.0, 3.0, 6.0))])
    s = make_solver(capi, p, len(X0))
    s.update_initialization(X0)
    opts = capi.default_opts(max_iter=max_iter, check_every=max_iter)
    for _ in range(3):
        s.debug_poison_lds()
        s.calculate(opts)
        r = s.get_results()
        assert np.all(r["status"] == 0) and np.isfinite(r["u"]).all() and np.isfinite(r["x"]).all()
    s.close()
    for i in range(0, len(X0), 7):
        assert np.abs(r["u"][i] - mo.solve_mpc_exact(p, X0[i])["u"]).max() <= U_TOL


def test_repeatability(capi, mo):
    """Two launches on the same inputs give bitwise identical results (no atomics, fixed reduction order)."""
    p = mo.quadrotor()
    X0 = mo.quadrotor_x0_batch(256, 3.0)
    a = step(capi, p, X0)
    b = step(capi, p, X0)
    for k in ("u", "x", "iters", "polish_iters"):
        assert np.array_equal(a[k], b[k])


def test_polish_through_l2_build_matches_lds_build(capi, mo, monkeypatch):
    """k_polish<false> (G read through L2, 4 waves per workgroup) is what shapes whose G does not fit LDS run; forced
    here on the benchmark shape (ALMPC_POLISH_NO_GLDS=1 is read by almpc_create) it must give the results of the
    default k_polish<true> build: same status, same iteration counts, u equal to rounding."""
    p = mo.quadrotor()
    X0 = np.concatenate([mo.quadrotor_x0_batch(64, a, first_instance=64 * k) for k, a in enumerate((0.3, 1.0, 3.0, 6.0))])
    def run():
        s = capi.Solver(p.n, p.m, p.N, len(X0))
        s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, rho=30.0, rho_profile="stiffness")
        s.set_reference(p.x_ref, p.u_ref)
        s.update_initialization(X0)
        s.calculate(capi.default_opts(rho=30.0, max_iter=8, check_every=8))
        r = s.get_results()
        s.close()
        return r

    a = run()
    monkeypatch.setenv("ALMPC_POLISH_NO_GLDS", "1")
    b = run()
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["polish_iters"], b["polish_iters"])
    assert np.abs(a["u"] - b["u"]).max() <= 1e-12 and np.abs(a["x"] - b["x"]).max() <= 1e-10
    ok = a["status"] == 0
    for i in np.nonzero(ok)[0][::7]:
        e = mo.solve_mpc_exact(p, X0[i])
        assert np.abs(a["u"][i] - e["u"]).max() <= U_TOL


def test_fused_step_kernel_equals_two_kernel_path(capi, mo):
    """k_step_fused (one kernel per step: ADMM phase + polish of the same tile) and the two-kernel path (k_admm, k_polish<true>) run the
    same device functions: identical status, iteration counts and results, cold and warm, also for a partial last tile."""
    p = mo.quadrotor()
    X0 = np.concatenate([mo.quadrotor_x0_batch(50, a, first_instance=70 * k) for k, a in enumerate((0.3, 1.0, 3.0, 6.0))])  # 200 = 12.5 tiles
    s = capi.Solver(p.n, p.m, p.N, len(X0))
    s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, rho=30.0, rho_profile="stiffness")
    s.set_reference(p.x_ref, p.u_ref)
    out = {}
    for fused in (True, False):
        s.set_step_fusion(fused)
        s.update_initialization(X0)
        s.calculate(capi.default_opts(rho=30.0, max_iter=8, check_every=8))
        a = s.get_results()
        s.update_initialization(X0 * 0.9)
        s.calculate(capi.default_opts(rho=30.0, max_iter=8, check_every=8, warm_start=1))
        out[fused] = (a, s.get_results())
    s.close()
    for k in (0, 1):
        f, t = out[True][k], out[False][k]
        for key in ("status", "iters", "polish_iters", "u", "x", "e_u", "e_x"):
            assert np.array_equal(f[key], t[key]), key
    e = mo.solve_mpc_exact(p, X0[7])
    assert np.abs(out[True][0]["u"][7] - e["u"]).max() <= U_TOL


def test_one_wave_step_of_small_shared_problems_equals_the_two_launch_path(capi, mo, qtp_ab):
    """Round 5 (review item 8): shared-model problems with nz <= 64 -- configs[0] (double integrator, nz 10), the reference's own test
    size (QTP fixture, N 5, m 2: test/computation_mpc_test.jl:981-1054) and a mid-size one (N 20: nz 40) -- run ONE kernel per step,
    one wave per instance (k_step_inst_wave on the shared operands).  Same statuses, same ADMM iteration counts and u within 1e-9 of
    the two-launch path (ALMPC_NO_SHARED_WAVE=1: k_admm's 16-instance MFMA tile + k_polish), cold and warm, every instance within
    1e-6 of the exact oracle; no-warm-state option and per-instance references included."""
    import os
    cases = []
    p = mo.double_integrator()
    cases.append((p, np.array([[1.0, 0.0], [3.0, -1.0], [-4.0, 2.0], [0.2, 0.1], [9.0, 0.0]])))
    A, B = qtp_ab
    pq = mo.qtp_linear_fixture_problem(A, B)
    cases.append((pq, 0.6 + 0.3 * mo.splitmix_normal(7, 0, 37, 4)))
    pm = mo.make_problem(A, B, 20, [0.0, 0.0], [4.0, 3.26], x_ref=0.65 * np.ones(4), u_ref=1.2 * np.ones(2))
    cases.append((pm, 0.65 + 0.8 * mo.splitmix_normal(8, 0, 70, 4)))
    for p, X0 in cases:
        b = len(X0)
        out = {}
        for env in ("0", "1"):
            os.environ["ALMPC_NO_SHARED_WAVE"] = env
            try:
                s = capi.Solver(p.n, p.m, p.N, b)
                s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max)
                s.set_reference(p.x_ref, p.u_ref)
                s.update_initialization(X0)
                s.calculate()
                a = s.get_results()
                s.update_initialization(X0 * 0.95)
                s.calculate(capi.default_opts(warm_start=1))
                w = s.get_results()
                s.update_initialization(X0)
                s.calculate(capi.default_opts(keep_warm_state=False))
                nw = s.get_results()
                uref_i = np.repeat(p.u_ref[None], b, 0) + 0.01 * np.arange(b)[:, None, None]
                s.set_reference(np.repeat(p.x_ref[None], b, 0), uref_i, per_instance=True)
                s.calculate()
                pi = s.get_results()
                s.close()
                out[env] = (a, w, nw, pi)
            finally:
                os.environ.pop("ALMPC_NO_SHARED_WAVE", None)
        for k in range(4):
            f, t = out["0"][k], out["1"][k]
            assert np.array_equal(f["status"], t["status"]) and np.all(f["status"] == 0), (p.nz, k)
            assert np.array_equal(f["iters"], t["iters"]), (p.nz, k)
            assert np.abs(f["u"] - t["u"]).max() <= 1e-9, (p.nz, k, np.abs(f["u"] - t["u"]).max())
            assert np.abs(f["x"] - t["x"]).max() <= 1e-8, (p.nz, k)
        for i in range(0, b, 3):
            e = mo.solve_mpc_exact(p, X0[i])
            assert np.abs(out["0"][0]["u"][i] - e["u"]).max() <= U_TOL, (p.nz, i)


def test_step_without_warm_state_gives_identical_results(capi, mo):
    """ALMPC_OPT_NO_WARM_STATE (opts.reserved[0]): the step skips the stores of the ADMM x and y (the polish gets the signs of y as
    flag words): bit-identical results on both kernel paths, and a warm start right after such a step is refused."""
    p = mo.quadrotor()
    X0 = np.concatenate([mo.quadrotor_x0_batch(50, a, first_instance=90 * k) for k, a in enumerate((0.3, 1.0, 3.0, 6.0))])  # partial last tile
    s = capi.Solver(p.n, p.m, p.N, len(X0))
    s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, rho=45.0, rho_profile="stiffness")
    s.set_reference(p.x_ref, p.u_ref)
    s.update_initialization(X0)
    out = {}
    for fused in (True, False):
        s.set_step_fusion(fused)
        for keep in (True, False):
            s.calculate(capi.default_opts(rho=45.0, max_iter=6, check_every=6, keep_warm_state=keep))
            out[fused, keep] = s.get_results()
    for key in ("status", "iters", "polish_iters", "u", "x", "e_u", "e_x"):
        for k in out:
            assert np.array_equal(out[k][key], out[True, True][key]), (k, key)
    with pytest.raises(capi.AlmpcError) as ei:   # the last step kept no state
        s.calculate(capi.default_opts(rho=45.0, max_iter=6, check_every=6, warm_start=1))
    assert ei.value.code == -1
    s.calculate(capi.default_opts(rho=45.0, max_iter=6, check_every=6))              # a step that keeps it ...
    s.calculate(capi.default_opts(rho=45.0, max_iter=6, check_every=6, warm_start=1))  # ... makes the warm start legal again
    assert np.all(s.get_results(want=("status",))["status"] == 0)
    s.close()
    assert np.all(out[True, True]["status"] == 0)


def test_config3_shards_reassemble_the_batch(pkg, capi, mo):
    """BASELINE configs[2] in miniature: a batch seeded with 0x5EED0003 split into 8 contiguous shards (sharding.shard_range, one
    handle per shard as one process per GPU would hold it) gives, shard by shard, exactly the results of the unsharded batch --
    instances never interact, there is no data-path collective."""
    p = mo.quadrotor()
    batch, world = 8 * 40 + 5, 8  # ragged: shards of 41 / 40 instances, none a multiple of the 16-instance tile
    X0 = mo.quadrotor_x0_batch(batch, 1.5, seed=0x5EED0003)
    whole = step(capi, p, X0)
    for rank in range(world):
        lo, hi = pkg.sharding.shard_range(batch, rank, world)
        part = step(capi, p, X0[lo:hi])
        for key in ("status", "iters", "polish_iters", "u", "x"):
            assert np.array_equal(part[key], whole[key][lo:hi]), (rank, key)
    assert np.all(whole["status"] == 0)
    assert np.abs(whole["u"][batch - 1] - mo.solve_mpc_exact(p, X0[batch - 1])["u"]).max() <= U_TOL
