"""CPU tests of the oracle: pinned against everything the reference's own tests hold for this path
(fixture data, scenario assertions, structural counts), cross-checked against independent scipy routes, and
frozen by the golden vectors.  Reference paths are relative to /root/reference."""
import json
import os

import numpy as np
import pytest
import scipy.linalg as sla

from conftest import GOLDEN


def _load(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def _problem(mo, g):
    return mo.make_problem(g["A"], g["B"], g["N"], g["u_min"], g["u_max"], x_ref=g["x_ref"], u_ref=g["u_ref"],
                           q=g["q"], r=g["r"], s=g["s"])


# ---------------------------------------------------------------------------- reference-held pins
def test_fixture_decode_matches_survey_values(qtp_ab):
    A, B = qtp_ab  # test/models_saved/linear_regressor_train_result.jls, read as in test/computation_mpc_test.jl:1003-1006
    assert A.shape == (4, 4) and B.shape == (4, 2)
    np.testing.assert_allclose(A[0], [0.968072, -0.00422939, 0.0207686, 0.00121679], rtol=5e-6)
    np.testing.assert_allclose(A[3], [-0.00778791, -0.0117474, 0.0042763, 0.984822], rtol=5e-6)
    np.testing.assert_allclose(B[:, 0], [0.00622003, 9.12009e-05, -0.000222272, 0.0144097], rtol=5e-6)
    np.testing.assert_allclose(B[:, 1], [4.15756e-05, 0.00806921, 0.0122243, -0.000295922], rtol=5e-6)


def test_reference_scenario_assertions(mo, qtp_ab):
    """The only numeric assertions the reference makes on this path (test/computation_mpc_test.jl:1053-1054):
    x ~ x_ref (atol 0.5), u[:,1] ~ u_ref (atol 3)."""
    p = mo.qtp_linear_fixture_problem(*qtp_ab)
    for sol in (mo.solve_mpc_exact(p, np.full(4, 0.6)), mo.solve_mpc_admm_polish(p, np.full(4, 0.6))):
        assert np.all(np.abs(sol["x"] - 0.65) <= 0.5)
        assert np.all(np.abs(sol["u"][:, 0] - 1.2) <= 3.0)
        # e_x = x - x_reference, e_u = u - u_reference (..linear.jl:81-87)
        np.testing.assert_allclose(sol["e_x"], sol["x"] - p.x_ref, atol=1e-15)
        np.testing.assert_allclose(sol["e_u"], sol["u"] - p.u_ref, atol=1e-15)
        np.testing.assert_allclose(sol["x"][:, 0], 0.6)  # x[:,1] is fixed to x0 (src/main/computation_mpc.jl:23-27)


def test_structural_constraint_counts(mo, qtp_ab):
    """test/terminal_ingredient_test.jl:160,317: 74 constraints for "none", 78 for "equality" at N=5, n=4, m=2
    (20 dynamics + 20 input-bound rows + 24 e_x defs + 10 e_u defs [+ 4])."""
    p = mo.qtp_linear_fixture_problem(*qtp_ab)
    assert mo.sparse_problem(p, np.full(4, 0.6))["n_constraints"] == 74
    p.terminal = "equality"
    assert mo.sparse_problem(p, np.full(4, 0.6))["n_constraints"] == 78
    # with the state box (kw mpc_state_constraint present): + 2*n*(N+1) rows (..linear.jl:62-70)
    p.terminal = "none"
    p.x_min, p.x_max = np.full(4, 0.2), np.array([1.36, 1.36, 1.30, 1.30])
    assert mo.sparse_problem(p, np.full(4, 0.6))["n_constraints"] == 74 + 2 * 4 * 6


def test_sparse_problem_size_config2(mo):
    """1,476 variables and 1,596 rows at n=12, m=4, N=30."""
    sp = mo.sparse_problem(mo.quadrotor(), np.zeros(12))
    assert sp["nv"] == 1476 and sp["A"].shape == (1596, 1476) and sp["n_constraints"] == 1092


# ---------------------------------------------------------------------------- independent cross-checks
@pytest.mark.parametrize("which", ["di", "qtp", "quad"])
def test_dare_matches_scipy(mo, qtp_ab, which):
    p = {"di": mo.double_integrator, "quad": mo.quadrotor}.get(which, lambda: mo.qtp_linear_fixture_problem(*qtp_ab))()
    P = sla.solve_discrete_are(p.A, p.B, p.Q, p.R)
    assert np.abs(p.P - P).max() <= 1e-9 * np.abs(P).max()
    assert np.abs(mo.dare_residual(p.A, p.B, p.Q, p.R, p.P)).max() <= 1e-8 * np.abs(P).max()


def test_survey_known_answers(mo, qtp_ab):
    """SURVEY.md Appendix C / section 8c(4) digits (scratch values of the survey, cross-check only)."""
    p = mo.double_integrator()
    np.testing.assert_allclose(p.P, [[200.0665853543, 50.0999001995], [50.0999001995, 125.1832094953]], rtol=1e-10)
    v = mo.solve_mpc_exact(p, np.array([1.0, 0.0]))["v"]
    np.testing.assert_allclose(v[:4], [-0.6660752235, 0.4433606950, 0.1485413291, 0.0494708317], atol=1e-9)
    v = mo.solve_mpc_exact(p, np.array([5.0, 0.0]))["v"]
    np.testing.assert_allclose(v[:4], [-1, -1, 0.6669622406, 0.8881993255], atol=1e-9)
    q = mo.qtp_linear_fixture_problem(*qtp_ab)
    s = mo.solve_mpc_exact(q, np.full(4, 0.6))
    np.testing.assert_allclose(s["u"][:, 0], [2.7559412659, 2.9550746591], atol=1e-8)
    np.testing.assert_allclose(s["u"][:, 4], [1.3904599197, 1.4432352234], atol=1e-8)
    np.testing.assert_allclose(s["x"][:, 5], [0.6266243663, 0.6385893605, 0.6545269524, 0.6543881816], atol=1e-8)
    assert abs(q.P[0, 0] - 1547.2440077) < 1e-5


@pytest.mark.parametrize("which,x0", [("di", [1.0, 0.0]), ("di", [5.0, 0.0]), ("qtp", [0.6] * 4)])
def test_condensed_equals_sparse_multiple_shooting(mo, qtp_ab, which, x0):
    """The condensed QP has the minimiser of the QP the reference poses (sparse, with explicit states and all six
    variable blocks): OSQP iteration restated on the sparse statement, run to a tight tolerance."""
    p = mo.double_integrator() if which == "di" else mo.qtp_linear_fixture_problem(*qtp_ab)
    x0 = np.array(x0)
    sp = mo.sparse_problem(p, x0)
    r = mo.osqp_admm(sp["P"], sp["q"], sp["A"], sp["l"], sp["u"], eps_abs=1e-10, eps_rel=1e-10, max_iter=40000)
    assert r["status"] == 0
    u_sparse = r["x"][sp["idx"]["u"]:sp["idx"]["u"] + p.m * p.N].reshape(p.N, p.m).T
    x_sparse = r["x"][sp["idx"]["x"]:sp["idx"]["x"] + p.n * (p.N + 1)].reshape(p.N + 1, p.n).T
    e = mo.solve_mpc_exact(p, x0)
    assert np.abs(u_sparse - e["u"]).max() <= 1e-6
    assert np.abs(x_sparse - e["x"]).max() <= 1e-6


def test_osqp_default_tolerance_is_loose(mo, qtp_ab):
    """What the reference's solver actually returns at OSQP defaults (eps 1e-3, 25-iteration checks) is far from
    the optimum -- consistent with the reference's own atol = 3 on u; this is why parity is defined on the exact
    optimum (SURVEY.md section 7 hard part 2)."""
    p = mo.qtp_linear_fixture_problem(*qtp_ab)
    sp = mo.sparse_problem(p, np.full(4, 0.6))
    r = mo.osqp_admm(sp["P"], sp["q"], sp["A"], sp["l"], sp["u"])
    u = r["x"][sp["idx"]["u"]:sp["idx"]["u"] + 10].reshape(5, 2).T
    assert r["status"] == 0 and np.abs(u[:, 0] - 1.2).max() <= 3.0
    assert np.abs(u - mo.solve_mpc_exact(p, np.full(4, 0.6))["u"]).max() > 1e-3


def test_exact_solver_vs_bvls(mo, qtp_ab):
    """Independent route: bounded least squares on the Cholesky-transformed problem (well-conditioned case)."""
    from scipy.optimize import lsq_linear
    p = mo.qtp_linear_fixture_problem(*qtp_ab)
    p.u_max = np.array([2.0, 2.5])  # make bounds active
    x0 = np.full(4, 0.45)
    H, f, lo, hi = mo.condensed_qp(p, x0)
    c = np.linalg.cholesky(H)
    r = lsq_linear(c.T, -np.linalg.solve(c, f), bounds=(lo, hi), method="bvls", tol=1e-14)
    v = mo.solve_box_qp_exact(H, f, lo, hi)
    assert ((v <= lo) | (v >= hi)).sum() >= 1
    assert np.abs(r.x - v).max() <= 1e-9


def test_kkt_certificate_detects_wrong_point(mo):
    p = mo.double_integrator()
    H, f, lo, hi = mo.condensed_qp(p, np.array([5.0, 0.0]))
    v = mo.solve_box_qp_exact(H, f, lo, hi)
    assert mo.kkt_residual(H, f, lo, hi, v) <= 1e-9
    assert mo.kkt_residual(H, f, lo, hi, v + 1e-3) > 1e-4


# ---------------------------------------------------------------------------- state rows (state box, terminal equality)
@pytest.mark.parametrize("case,x0", [("box", [5.0, 0.0]), ("box", [-6.0, 0.5]), ("box", [1.0, 0.0]), ("eq", [2.0, 0.0]),
                                     ("eq", [1.0, -0.5]), ("box_eq", [3.0, 0.5])])
def test_state_rows_exact_equals_sparse_statement(mo, case, x0):
    """Dual active set in constraint space vs the reference's own sparse statement (with its state-box rows,
    ..linear.jl:62-70, and terminal equality rows, src/sub/design_mpc.jl:330-331) solved by the OSQP restatement."""
    kw = dict(box=dict(x_min=[-10.0, -0.8], x_max=[10.0, 0.8]), eq=dict(terminal="equality"),
              box_eq=dict(x_min=[-10.0, -0.8], x_max=[10.0, 0.8], terminal="equality"))[case]
    p = mo.make_problem([[1.0, 1.0], [0.0, 1.0]], [[0.5], [1.0]], 10, [-1.0], [1.0], **kw)
    x0 = np.array(x0)
    e = mo.solve_mpc_exact(p, x0, return_info=True)
    sp = mo.sparse_problem(p, x0)
    r = mo.osqp_admm(sp["P"], sp["q"], sp["A"], sp["l"], sp["u"], eps_abs=1e-10, eps_rel=1e-10, max_iter=200000)
    assert r["status"] == 0
    u_sparse = r["x"][sp["idx"]["u"]:sp["idx"]["u"] + 10].reshape(10, 1).T
    assert np.abs(u_sparse - e["u"]).max() <= 1e-6
    if case != "eq":
        assert np.all(e["x"][1] >= -0.8 - 1e-10) and np.all(e["x"][1] <= 0.8 + 1e-10)
    if case != "box":
        assert np.abs(e["e_x"][:, -1]).max() <= 1e-10


def test_state_rows_infeasible_cases(mo):
    p = mo.make_problem([[1.0, 1.0], [0.0, 1.0]], [[0.5], [1.0]], 10, [-1.0], [1.0], x_min=[-10.0, -0.8], x_max=[10.0, 0.8])
    with pytest.raises(ValueError):
        mo.solve_mpc_exact(p, np.array([20.0, 0.0]))   # x0 outside the box: stage 1 is constrained too
    with pytest.raises(ValueError):
        mo.solve_mpc_exact(p, np.array([9.9, 0.8]))    # inside now, but cannot stay inside
    pe = mo.make_problem([[1.0, 1.0], [0.0, 1.0]], [[0.5], [1.0]], 10, [-1.0], [1.0], terminal="equality")
    with pytest.raises(ValueError):
        mo.solve_mpc_exact(pe, np.array([30.0, 0.0]))  # origin not reachable in N steps with |u| <= 1


def test_state_box_inactive_equals_box_only(mo):
    q = mo.quadrotor()
    qb = mo.make_problem(q.A, q.B, 30, q.u_min, q.u_max, x_min=-1e3 * np.ones(12), x_max=1e3 * np.ones(12))
    x0 = mo.quadrotor_x0_batch(1, 3.0)[0]
    np.testing.assert_allclose(mo.solve_mpc_exact(qb, x0)["u"], mo.solve_mpc_exact(q, x0)["u"], atol=1e-8)


def test_fnn_jacobian_matches_finite_differences(mo):
    for act in ("relu", "identity", "tanh", "sigmoid", "swish"):
        f = mo.synthetic_fnn(act=act)
        x, u = np.array([0.3, -0.2, 0.1, 0.4]), np.array([0.2, -0.5])
        A, B = f.jacobian(x, u)
        eps = 1e-6
        Afd = np.array([(f.forward(x + eps * np.eye(4)[j], u) - f.forward(x - eps * np.eye(4)[j], u)) / (2 * eps) for j in range(4)]).T
        Bfd = np.array([(f.forward(x, u + eps * np.eye(2)[j]) - f.forward(x, u - eps * np.eye(2)[j])) / (2 * eps) for j in range(2)]).T
        assert np.abs(A - Afd).max() <= 1e-8 and np.abs(B - Bfd).max() <= 1e-8
    A0, _ = mo.synthetic_fnn().jacobian(np.zeros(4), np.zeros(2))
    assert abs(np.max(np.abs(np.linalg.eigvals(A0))) - 0.95) < 1e-9


# ---------------------------------------------------------------------------- golden vectors
@pytest.mark.parametrize("name", ["double_integrator", "double_integrator_S", "qtp_linear", "quadrotor"])
def test_golden_vectors(mo, name):
    g = _load(name)
    p = _problem(mo, g)
    np.testing.assert_allclose(p.P, g["P"], rtol=1e-9)
    des = mo.design_shared(p)
    for c in g["cases"]:
        x0 = np.array(c["x0"])
        e = mo.solve_mpc_exact(p, x0)
        assert np.abs(e["u"] - np.array(c["u"])).max() <= 1e-9
        assert np.abs(e["x"] - np.array(c["x"])).max() <= 1e-7
        a = mo.solve_mpc_admm_polish(p, x0, des, max_iter=50)  # the algorithm of the HIP path
        assert np.abs(a["u"] - np.array(c["u"])).max() <= 1e-7


def test_golden_has_active_bounds(mo):
    g = _load("quadrotor")
    na = [c["n_active"] for c in g["cases"]]
    assert max(na[:6]) <= 2 and max(na[12:18]) >= 5 and max(na[18:]) > 32  # amplitude classes 0.3 / 3 / 10


# ---------------------------------------------------------------------------- C restatement vs numpy restatement
@pytest.mark.parametrize("amp", [0.3, 1.0, 3.0, 10.0])
def test_c_oracle_matches_numpy(mo, co, amp):
    p = mo.quadrotor()
    des = mo.design_shared(p)
    X0 = mo.quadrotor_x0_batch(12, amp, first_instance=7)
    r = co.step_batch(p, des, X0, max_iter=50, threads=2)
    for i in range(len(X0)):
        fs = des["Fs"] @ X0[i]
        a = mo.admm_box(des["Hs"], fs, des["lo"], des["hi"], Minv=des["Minv"], unscale=des["d"], max_iter=50)
        assert a["iters"] == r["iters"][i]
        pol = mo.polish_active_set(des["G"], -des["G"] @ fs, des["lo"], des["hi"], a["z"], a["y"], refine=False)
        assert pol["iters"] == r["polish_iters"][i]
        e = mo.solve_mpc_exact(p, X0[i])
        assert np.abs(r["u"][i] - e["u"]).max() <= 1e-7
        assert np.abs(r["x"][i] - e["x"]).max() <= 1e-6
        assert r["status"][i] == 0


def test_c_oracle_no_polish_is_admm_iterate(mo, co):
    p = mo.double_integrator()
    des = mo.design_shared(p)
    X0 = np.array([[5.0, 0.0]])
    r = co.step_batch(p, des, X0, polish=False, max_iter=4000)
    a = mo.admm_box(des["Hs"], des["Fs"] @ X0[0], des["lo"], des["hi"], Minv=des["Minv"], unscale=des["d"])
    np.testing.assert_allclose(r["e_u"][0].T.reshape(-1), a["z"] * des["d"], atol=1e-12)
    assert r["iters"][0] == a["iters"] and r["status"][0] == a["status"] == 0


# ---------------------------------------------------------------------------- synthetic input generator
def test_generator_is_frozen_and_shardable(mo):
    z = mo.splitmix_normal(0x5EED0002, 0, 4, 12)
    assert z.shape == (4, 12) and np.isfinite(z).all()
    np.testing.assert_allclose(z, mo.splitmix_normal(0x5EED0002, 0, 4, 12))
    np.testing.assert_array_equal(mo.splitmix_normal(0x5EED0002, 2, 2, 12), z[2:])  # stream = instance index
    big = mo.splitmix_normal(0x5EED0002, 0, 20000, 12)
    assert abs(big.mean()) < 0.01 and abs(big.std() - 1.0) < 0.01
    assert np.abs(np.corrcoef(big[:, 0], big[:, 1])[0, 1]) < 0.03
    assert not np.allclose(mo.splitmix_normal(0x5EED0003, 0, 4, 12), z)


def test_quadrotor_model_is_exact_zoh(mo):
    A, B = mo.quadrotor_model()
    # position integrates velocity exactly; thrust acts on vz with 1/m
    np.testing.assert_allclose(A[0, 3], 0.1)
    np.testing.assert_allclose(B[5, 0], 0.1 / 0.5)
    np.testing.assert_allclose(B[2, 0], 0.5 * 0.1 ** 2 / 0.5)
    # torque -> angle -> velocity -> position chain: tau_y moves x by g*Ts^4/(24 J)
    np.testing.assert_allclose(B[0, 2], 9.81 * 0.1 ** 4 / 24 / 4e-3, rtol=1e-12)


# ---------------------------------------------------------------------------- NLP branch (SQP restatement)
def test_nlp_gradient_matches_finite_differences(mo):
    f = mo.synthetic_fnn(act="tanh")
    n, m, N = 4, 2, 6
    x_ref = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N + 1)); u_ref = np.tile(np.array([0.1, -0.2])[:, None], (1, N))
    Q, R, S, P = 100.0 * np.eye(n), 0.1 * np.eye(m), 0.3 * np.eye(m), 150.0 * np.eye(n)
    x0 = np.array([0.5, -0.3, 0.2, 0.1])
    U = 0.3 * mo.splitmix_normal(0x5EED0006, 0, m, N)
    J, G, _ = mo.nlp_cost_and_gradient(f, x0, U, x_ref, u_ref, Q, R, S, P)
    eps = 1e-6
    for a in range(m):
        for k in range(N):
            Up, Um = U.copy(), U.copy()
            Up[a, k] += eps; Um[a, k] -= eps
            fd = (mo.nlp_cost_and_gradient(f, x0, Up, x_ref, u_ref, Q, R, S, P)[0] - mo.nlp_cost_and_gradient(f, x0, Um, x_ref, u_ref, Q, R, S, P)[0]) / (2 * eps)
            assert abs(fd - G[a, k]) <= 1e-6 * max(1.0, abs(G[a, k]))


def test_sqp_restatement_reaches_a_kkt_point_of_the_nlp(mo):
    f = mo.synthetic_fnn(act="tanh")
    n, m, N = 4, 2, 12
    x_ref = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N + 1)); u_ref = np.tile(np.array([0.1, -0.2])[:, None], (1, N))
    Q, R, S, P = 100.0 * np.eye(n), 0.1 * np.eye(m), 0.0 * np.eye(m), 150.0 * np.eye(n)
    umin, umax = -np.ones(m), np.ones(m)
    x0 = x_ref[:, 0] + np.array([0.5, -0.4, 0.3, 0.2])
    X, U, hist = mo.sqp_fnn(f, x0, x_ref, u_ref, Q, R, S, P, umin, umax, iters=40)
    assert hist[-1][0] <= 1e-9 and hist[-1][1] <= 1e-12, hist[-3:]
    assert np.abs(X - mo.fnn_rollout(f, x0, U)).max() <= 1e-10          # multiple shooting closed the defects
    assert mo.nlp_kkt_residual(f, x0, U, x_ref, u_ref, Q, R, S, P, umin, umax) <= 1e-8
    assert np.any(U >= 1 - 1e-12) or np.any(U <= -1 + 1e-12)            # the box is active somewhere
    # and it is a minimiser, not just a stationary point: random feasible perturbations do not decrease the cost
    J0 = mo.nlp_cost_and_gradient(f, x0, U, x_ref, u_ref, Q, R, S, P)[0]
    for s in range(5):
        Up = np.clip(U + 1e-3 * mo.splitmix_normal(0x5EED0007, s, m, N), umin[:, None], umax[:, None])
        assert mo.nlp_cost_and_gradient(f, x0, Up, x_ref, u_ref, Q, R, S, P)[0] >= J0 - 1e-12


def _sqp_golden(mo):
    g = _load("fnn_sqp")
    f = mo.FnnModel(np.array(g["W_in"]), [np.array(w) for w in g["W_h"]], [np.array(b) for b in g["b_h"]], np.array(g["W_out"]), g["act"])
    n, m, N = g["n"], g["m"], g["N"]
    kw = dict(x_ref=np.tile(np.array(g["x_ref"])[:, None], (1, N + 1)), u_ref=np.tile(np.array(g["u_ref"])[:, None], (1, N)),
              Q=g["q"] * np.eye(n), R=g["r"] * np.eye(m), S=g["s"] * np.eye(m), P=np.array(g["P"]),
              u_min=np.array(g["u_min"]), u_max=np.array(g["u_max"]))
    return g, f, kw


def test_golden_sqp_vectors(mo):
    """The committed NLP-branch vectors: the stored inputs are certified KKT points of the stored problem (independent of how
    they were found), and the SQP restatement reproduces them."""
    g, f, kw = _sqp_golden(mo)
    f0 = mo.synthetic_fnn(act="tanh")
    assert np.abs(f.W_out - f0.W_out).max() == 0.0 and np.abs(f.W_h[1] - f0.W_h[1]).max() == 0.0   # same generator as the GPU tests use
    for c in g["cases"]:
        x0, U = np.array(c["x0"]), np.array(c["u"])
        assert mo.nlp_kkt_residual(f, x0, U, **kw) <= 1e-9
        assert np.abs(np.array(c["x"]) - mo.fnn_rollout(f, x0, U)).max() <= 1e-10
        assert c["n_active"] >= 10
    c = g["cases"][1]
    X, U, _ = mo.sqp_fnn(f, np.array(c["x0"]), kw["x_ref"], kw["u_ref"], kw["Q"], kw["R"], kw["S"], kw["P"], kw["u_min"], kw["u_max"], iters=60)
    assert np.abs(U - np.array(c["u"])).max() <= 1e-9


def test_ltv_qp_reduces_to_the_condensed_qp_and_to_finite_differences(mo):
    """ltv_qp (the QP of one SQP iteration) pinned from two sides: with time-invariant stages, zero defects and the linearisation
    trajectory generated by the same model it is the condensed QP of the linear path shifted to v = u - ubar (same minimiser u);
    and for a nonlinear model its gradient at v = 0 is the exact gradient of the NLP cost along the linearised dynamics."""
    # (the reference pair must be an equilibrium of the model: the reference's deviation dynamics drop f(x_ref, u_ref) - x_ref,
    # SURVEY.md section 8a-1, the LTV statement keeps the true dynamics)
    p = mo.make_problem([[1.0, 1.0], [0.0, 1.0]], [[0.5], [1.0]], 8, [-1.0], [1.0], x_ref=[0.3, 0.0], u_ref=[0.0], s=0.7)
    x0 = np.array([2.0, -0.5])
    rng = np.random.default_rng(5)
    ubar = np.clip(0.3 * rng.standard_normal((1, p.N)), -1, 1)
    xbar = np.empty((2, p.N + 1)); xbar[:, 0] = x0
    for k in range(p.N):
        xbar[:, k + 1] = p.A @ xbar[:, k] + p.B @ ubar[:, k]
    H, q, lo, hi = mo.ltv_qp([p.A] * p.N, [p.B] * p.N, None, xbar, ubar, p.x_ref, p.u_ref, p.Q, p.R, p.S, p.P, p.u_min, p.u_max)
    v = mo.solve_box_qp_exact(H, q, lo, hi)
    e = mo.solve_mpc_exact(p, x0)
    assert np.abs((ubar + v.reshape(p.N, 1).T) - e["u"]).max() <= 1e-9
    # nonlinear model: q = dJ/du of the single-shooting cost at ubar when the trajectory is the model's own rollout
    f = mo.synthetic_fnn(act="tanh")
    n, m, N = 4, 2, 7
    x_ref = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N + 1)); u_ref = np.tile(np.array([0.1, -0.2])[:, None], (1, N))
    Q, R, S, P = 100.0 * np.eye(n), 0.1 * np.eye(m), 0.3 * np.eye(m), 150.0 * np.eye(n)
    U = 0.3 * mo.splitmix_normal(0x5EED000B, 0, m, N)
    x0 = np.array([0.5, -0.3, 0.2, 0.1])
    X = mo.fnn_rollout(f, x0, U)
    A, B = zip(*[f.jacobian(X[:, k], U[:, k]) for k in range(N)])
    H, q, lo, hi = mo.ltv_qp(list(A), list(B), None, X, U, x_ref, u_ref, Q, R, S, P, -np.ones(m), np.ones(m))
    _, G, _ = mo.nlp_cost_and_gradient(f, x0, U, x_ref, u_ref, Q, R, S, P)
    assert np.abs(q - G.T.reshape(-1)).max() <= 1e-9 * max(1.0, np.abs(q).max())
    assert np.allclose(H, H.T) and np.linalg.eigvalsh(H).min() > 0


# ---------------------------------------------------------------------------- reference-held Fnn fixture
def _fnn_fixture(mo):
    import json
    with open(os.path.join(GOLDEN, "fnn_qtp_fixture.json")) as f:
        g = json.load(f)
    model = mo.FnnModel(np.array(g["W_in"]), [np.array(w) for w in g["W_h"]], [np.array(b) for b in g["b_h"]], np.array(g["W_out"]),
                        g["activation"])
    sc = g["scenario"]
    N = sc["horizon"]
    xr = np.tile(np.array(sc["x_ref"])[:, None], (1, N + 1))
    ur = np.tile(np.array(sc["u_ref"])[:, None], (1, N))
    return g, model, sc, N, xr, ur


def test_fnn_fixture_decodes_to_a_plausible_qtp_model(mo, qtp_ab):
    """tests/golden/fnn_qtp_fixture.json = the chain of test/models_saved/fnn_train_result.jls (decoded by make_fnn_fixture.py): shapes
    of the reference's Fnn layout, and a model of the same quadruple-tank process as the reference's linear fixture -- its Jacobians at
    the test's reference point are close to that (A, B), and the reference point is close to a fixed point."""
    g, model, sc, N, xr, ur = _fnn_fixture(mo)
    assert (g["n"], g["m"], g["H"], g["L"]) == (4, 2, 13, 1)
    assert model.W_in.shape == (13, 6) and model.W_h[0].shape == (13, 13) and model.b_h[0].shape == (13,) and model.W_out.shape == (4, 13)
    np.testing.assert_allclose(model.forward(xr[:, 0], ur[:, 0]), g["check"]["f_at_reference"], rtol=1e-12)
    assert np.abs(model.forward(xr[:, 0], ur[:, 0]) - xr[:, 0]).max() <= 0.1
    A, B = model.jacobian(xr[:, 0], ur[:, 0])
    Al, Bl = qtp_ab
    assert np.abs(A - Al).max() <= 0.06 and np.abs(B - Bl).max() <= 0.03
    # float32 values widened exactly
    assert np.array_equal(model.W_in, model.W_in.astype(np.float32).astype(np.float64))


def test_fnn_fixture_reference_assertions_lp_vs_nlp(mo):
    """The reference's own test on this fixture (test/computation_mpc_test.jl:35-170): controllers of the LinearProgramming and the
    NonLinearProgramming branch from x0 = 0.6 towards x_ref = 0.65, u_ref = 1.2, horizon 5, and
        C_fnn_linear.x ~ C_fnn_nl.x atol 0.5 (:152),  e_x likewise (:163);
    the comparisons of u[:,1] (atol 0.1, :155) are marked `broken = true` there -- and are broken here too (the two optima differ by
    more than 2 in u[:,1]), which is as close to a reference-pinned number as this path gets."""
    g, model, sc, N, xr, ur = _fnn_fixture(mo)
    lo, hi, x0 = np.array(sc["u_low"]), np.array(sc["u_high"]), np.array(sc["x0"])
    p = mo.fnn_linear_problem(model, N, lo, hi, xr, ur)
    lin = mo.solve_mpc_exact(p, x0)
    X, U, hist = mo.sqp_fnn(model, x0, xr, ur, p.Q, p.R, p.S, p.P, lo, hi, 12, adaptive=True)
    assert mo.nlp_kkt_residual(model, x0, U, xr, ur, p.Q, p.R, p.S, p.P, lo, hi) <= 1e-9
    assert np.abs(lin["x"] - X).max() <= 0.5 and np.abs(lin["e_x"] - (X - xr)).max() <= 0.5     # the reference's assertions
    assert np.abs(lin["x"] - 0.65).max() <= 0.5                                                    # as :1053 asserts for the linear fixture
    assert np.abs(lin["u"][:, 0] - U[:, 0]).max() > 0.1                                            # `broken = true` in the reference
    assert np.all(U >= lo[:, None] - 1e-12) and np.all(U <= hi[:, None] + 1e-12)


# ---------------------------------------------------------------------------- structured (Riccati) restatement
def test_riccati_active_set_equals_the_exact_condensed_solver(mo):
    """mpc_oracle.riccati_active_set (the restatement k_riccati follows) solves the reference's QP in its multiple-shooting form;
    where the condensed problem is well conditioned it must land on the exact condensed optimum: double integrator, the QTP scenario
    shape, the benchmark plant at N = 30 and at N = 50 (m N = 200, beyond the condensed kernels), from both kinds of start."""
    p = mo.double_integrator()
    for x0 in ([1.0, 0.0], [5.0, 0.0], [-8.0, 3.0]):
        e = mo.solve_mpc_exact(p, np.array(x0))
        s = mo.solve_mpc_structured(p, np.array(x0))
        assert s["status"] == 0 and np.abs(s["u"] - e["u"]).max() <= 1e-9 and np.abs(s["x"] - e["x"]).max() <= 1e-8
    for N in (30, 50):
        p = mo.quadrotor(N=N)
        for amp, first in ((0.3, 0), (1.0, 11), (3.0, 23)):
            x0 = mo.quadrotor_x0_batch(1, amp, first_instance=first)[0]
            e = mo.solve_mpc_exact(p, x0)
            s = mo.solve_mpc_structured(p, x0)
            assert s["status"] == 0 and np.abs(s["u"] - e["u"]).max() <= 1e-8
            s2 = mo.solve_mpc_structured(p, x0, u_guess=np.clip(e["u"] + 0.01, p.u_min[:, None], p.u_max[:, None]))   # a warm guess
            assert s2["status"] == 0 and np.abs(s2["u"] - e["u"]).max() <= 1e-8 and s2["iters"] <= s["iters"] + 8


def test_riccati_active_set_on_an_open_loop_unstable_model(mo):
    """Spectral radius 2.2 over 20 stages: cond of the condensed Hessian ~1e14.  The structured solve does not notice; the optimum is
    certified by the KKT conditions of the reference's sparse statement (dynamics as constraints): zero defects, and the bound
    multipliers from the adjoint recursion have the right signs (recomputed here independently)."""
    rng = np.random.default_rng(3)
    n, m, N = 4, 2, 20
    A = rng.standard_normal((n, n)); A *= 2.2 / np.max(np.abs(np.linalg.eigvals(A)))
    B = rng.standard_normal((n, m))
    p = mo.make_problem(A, B, N, [-1, -1], [1, 1], P=200.0 * np.eye(n))
    x0 = 0.05 * rng.standard_normal(n)
    s = mo.solve_mpc_structured(p, x0)
    assert s["status"] == 0
    ex, eu = s["e_x"], s["e_u"]
    assert np.abs(A @ ex[:, :-1] + B @ eu - ex[:, 1:]).max() <= 1e-9 * max(1.0, np.abs(ex).max())
    lam = p.P @ ex[:, N]
    for k in range(N - 1, -1, -1):
        mu = 2.0 * (p.R @ eu[:, k] + B.T @ lam)
        for a in range(m):
            if eu[a, k] >= 1.0 - 1e-12:
                assert mu[a] <= 1e-7 * max(1.0, np.abs(mu).max())
            elif eu[a, k] <= -1.0 + 1e-12:
                assert mu[a] >= -1e-7 * max(1.0, np.abs(mu).max())
            else:
                assert abs(mu[a]) <= 1e-6 * max(1.0, np.abs(lam).max())
        lam = p.Q @ ex[:, k] + A.T @ lam


def test_riccati_active_set_solves_the_qp_of_an_sqp_iteration(mo):
    """Time-varying stage models, defects, state errors of the linearisation point and the input gradient (what k_riccati gets from the
    SQP loop): the stage-wise solve equals the exact solve of the condensed LTV QP (mpc_oracle.ltv_qp), iteration by iteration."""
    f = mo.synthetic_fnn(act="tanh")
    n, m, N = 4, 2, 20
    x_ref = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N + 1)); u_ref = np.tile(np.array([0.1, -0.2])[:, None], (1, N))
    Q, R, P, S = 100.0 * np.eye(n), 0.1 * np.eye(m), 150.0 * np.eye(n), np.zeros((m, m))
    x0 = x_ref[:, 0] + 0.6 * mo.splitmix_normal(0x5EED0005, 3, 1, n)[0]
    for it in (1, 3, 8):
        Xa, Ua, ha = mo.sqp_fnn(f, x0, x_ref, u_ref, Q, R, S, P, -np.ones(m), np.ones(m), it)
        Xb, Ub, hb = mo.sqp_fnn(f, x0, x_ref, u_ref, Q, R, S, P, -np.ones(m), np.ones(m), it, structured=True)
        assert np.abs(Ua - Ub).max() <= 1e-10 and np.abs(Xa - Xb).max() <= 1e-10
        assert abs(ha[-1][0] - hb[-1][0]) <= 1e-10 * max(1.0, ha[-1][0])


def test_ltv_state_rows_reduce_to_the_lti_state_box(mo):
    """ltv_qp + ltv_state_rows + solve_qp_rows_exact with A_k = A, B_k = B, zero inputs and the free response as linearisation
    trajectory is the state-box problem of solve_mpc_exact (same rows, same optimum); and an infeasible box raises."""
    p = mo.make_problem(np.array([[1.0, 1.0], [0.0, 1.0]]), np.array([[0.5], [1.0]]), 10, [-1.0], [1.0],
                        x_min=[-10.0, -0.8], x_max=[10.0, 0.8], terminal="equality")
    x0 = np.array([4.0, 0.0])
    e = mo.solve_mpc_exact(p, x0, return_info=True)
    assert e["info"]["n_active_state"] > 2   # (the two equality rows and velocity-box rows)
    N, n, m = p.N, 2, 1
    xbar = np.zeros((n, N + 1)); xbar[:, 0] = x0
    for k in range(N):
        xbar[:, k + 1] = p.A @ xbar[:, k]
    ubar = np.zeros((m, N))
    H, q, lo, hi, Gam, g = mo.ltv_qp([p.A] * N, [p.B] * N, None, xbar, ubar, p.x_ref, p.u_ref, p.Q, p.R, p.S, p.P, p.u_min, p.u_max,
                                     return_prediction=True)
    C, a0, lo_c, hi_c, eq_c = mo.ltv_state_rows(Gam, g, xbar, p.x_ref, p.x_min, p.x_max, "equality")
    v, W = mo.solve_qp_rows_exact(H, q, lo, hi, C, a0, lo_c, hi_c, eq_c)
    assert np.abs(v.reshape(N, m).T - e["u"]).max() <= 1e-8
    with pytest.raises(ValueError):
        C2, a2, l2, h2, e2 = mo.ltv_state_rows(Gam, g, xbar, p.x_ref, np.array([-10.0, -0.05]), np.array([10.0, 0.05]), "equality")
        mo.solve_qp_rows_exact(H, q, lo, hi, C2, a2, l2, h2, e2)


def test_feasibility_certificate_by_linear_programming(mo):
    """feasibility_slack: the LP certificate the full-batch state-row test classifies uncertified instances with.  Double integrator
    with a state box and the terminal equality: an interior start has a margin, a start the box cannot hold is infeasible by the
    overshoot, and a terminal equality out of reach is +inf."""
    p = mo.make_problem(np.array([[1.0, 1.0], [0.0, 1.0]]), np.array([[0.5], [1.0]]), 10, np.array([-1.0]), np.array([1.0]),
                        x_min=np.array([-5.0, -5.0]), x_max=np.array([5.0, 5.0]), terminal="equality")
    assert mo.feasibility_slack(p, np.array([1.0, 0.0])) < -1.0
    assert mo.feasibility_slack(p, np.array([4.9, 4.0])) == float("inf")      # cannot stop in 10 steps with |u| <= 1 from v = 4 ... and
    pb = mo.make_problem(p.A, p.B, 10, p.u_min, p.u_max, x_min=p.x_min, x_max=p.x_max)   # ... without the equality: overshoots the box
    t = mo.feasibility_slack(pb, np.array([4.9, 4.0]))
    assert 1.0 < t < 20.0
    # consistent with the exact solver: feasible <=> solve_mpc_exact returns, infeasible <=> it raises
    e = mo.solve_mpc_exact(p, np.array([1.0, 0.0]))
    assert np.abs(e["x"][:, -1]).max() <= 1e-9
    with pytest.raises((ValueError, RuntimeError)):
        mo.solve_mpc_exact(pb, np.array([4.9, 4.0]))


def test_saturated_lqr_seed_is_a_superset_guess_and_the_finish_does_not_depend_on_the_seed(mo):
    """Round-4 review item 1 (tools/exp_lqr_seed.py): the saturated closed-loop LQR rollout as the finish's seed.  The optimum
    must not depend on the seed; on amplitude-1 quadrotor states the seed is already the optimal working set (zero changes),
    and ADMM's sign(y) guess is a subset of the final set while the LQR seed is a superset of it."""
    p = mo.quadrotor()
    des = mo.design_shared(p, rho=45.0, rho_profile="stiffness")
    zero_changes = 0
    for i, amp in ((0, 1.0), (1, 1.0), (2, 1.0), (3, 3.0), (4, 3.0), (5, 3.0)):
        x0 = mo.quadrotor_x0_batch(1, amp, first_instance=i)[0]
        fs = des["Fs"] @ (x0 - p.x_ref[:, 0]) + des["fS"]
        v0 = -des["G"] @ fs
        r = mo.admm_box(des["Hs"], fs, des["lo"], des["hi"], rho=des["rho_vec"], sigma=des["sigma"], max_iter=6, check_every=6,
                        Minv=des["Minv"], unscale=des["d"])
        a = mo.polish_active_set(des["G"], v0, des["lo"], des["hi"], r["z"], r["y"])
        v, side = mo.saturated_lqr_rollout(p, x0)
        assert np.all(v >= np.tile(p.u_min, p.N) - 1e-15) and np.all(v <= np.tile(p.u_max, p.N) + 1e-15)
        b = mo.polish_active_set(des["G"], v0, des["lo"], des["hi"], v / des["d"], None, seed=side)
        assert np.max(np.abs(a["w"] - b["w"])) < 1e-7
        assert np.array_equal(a["side"], b["side"])
        exact = mo.solve_mpc_exact(p, x0)
        assert np.max(np.abs(b["w"] * des["d"] - (exact["u"] - p.u_ref).T.reshape(-1))) < 1e-6
        if amp == 1.0:
            zero_changes += int(b["n_add"] + b["n_remove"] + b["n_purged"] == 0)
            assert np.all((side != 0) | (a["side"] == 0))      # final set inside the LQR seed
    assert zero_changes == 3


def test_reachability_screen_never_contradicts_the_phase1_lp(mo):
    """The interval screen of the state box (k_state_box_screen; round 5) is a necessary condition of feasibility: whatever it rejects
    the phase-1 LP rejects too, and on the benchmark's clipped states it finds most of the infeasible instances at stage 1."""
    xbox = np.array([3, 3, 3, 1.5, 1.5, 1.5, 0.3, 0.3, 0.3, 1.0, 1.0, 1.0])
    q = mo.quadrotor(20)
    p = mo.make_problem(q.A, q.B, 20, q.u_min, q.u_max, x_min=-xbox, x_max=xbox)
    amp = np.array([0.3, 1.0, 3.0])
    caught = lp_inf = 0
    for i in range(60):
        x0 = np.clip(mo.quadrotor_x0_batch(1, amp[i % 3], first_instance=i)[0], -0.99 * xbox, 0.99 * xbox)
        k = mo.reachability_screen(p, x0)
        inf = mo.feasibility_slack(p, x0) > 1e-9
        assert not (k and not inf), (i, k)        # sound
        caught += bool(k)
        lp_inf += bool(inf)
    assert lp_inf >= 8 and caught >= lp_inf - 3     # incomplete, but close on this workload
    # a time-varying input reference and a non-zero state reference go through the same rule
    pr = mo.make_problem(q.A, q.B, 20, q.u_min, q.u_max, x_min=-xbox, x_max=xbox, x_ref=0.2 * np.ones(12),
                         u_ref=0.01 * np.sin(np.arange(20))[None, :] * np.ones((4, 1)))
    for i in range(12):
        x0 = np.clip(mo.quadrotor_x0_batch(1, 3.0, first_instance=100 + i)[0], -0.99 * xbox, 0.99 * xbox)
        if mo.reachability_screen(pr, x0):
            assert mo.feasibility_slack(pr, x0) > 1e-9, i
