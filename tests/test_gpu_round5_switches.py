"""Round-5 mechanisms against their own A/B switches (pytest -m gpu): each one changes HOW a result is reached, never the result.
  * the reachability screen of the state box in front of the stage-wise solve (ALMPC_SDUAL_NO_SCREEN) -- reference: the state box on
    every stage, .../linear/mpc_modeler_implementation_linear.jl:62-70; every screened verdict is also held against the phase-1 LP;
  * the capacity tiers of k_sdual handing over the inverse of their working set (ALMPC_SDUAL_NO_SINV_HANDOVER);
  * the stage-wise redo starting from the working set the state-row finish gave up with (ALMPC_NO_REDO_START);
  * the start of that redo with its inverse built in registers by k_sdual_start (ALMPC_SDUAL_NO_START_BUILD);
  * the redo enqueued gated behind the step BEFORE a synchronous look when the previous look found work (ALMPC_NO_PREDICTED_REDO);
  * the packed-triangle KKT inverse of k_admm_inst (ALMPC_NO_PACKED_MINV), even and odd nz.
The switches are read with getenv at call time, so one process can run both sides."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

XBOX = np.array([3, 3, 3, 1.5, 1.5, 1.5, 0.3, 0.3, 0.3, 1.0, 1.0, 1.0])


def _both(env, fn):
    out = {}
    for v in ("0", "1"):
        if v == "1":
            os.environ[env] = "1"
        try:
            out[v] = fn()
        finally:
            os.environ.pop(env, None)
    return out["0"], out["1"]


def _mixed_x0(mo, batch, first=0):
    amp = np.array([0.3, 1.0, 3.0])[np.arange(first, first + batch) % 3]
    return mo.splitmix_normal(0x5EED0002, first, batch, 12) * mo.QUADROTOR_X0_SCALE[None] * amp[:, None]


def test_reachability_screen_changes_no_verdict_and_is_certified_by_the_lp(capi, mo):
    N, batch = 50, 1536
    p = mo.make_problem(*mo.quadrotor_model(), N, mo.quadrotor().u_min, mo.quadrotor().u_max, x_min=-XBOX, x_max=XBOX)
    X0 = np.clip(_mixed_x0(mo, batch), -0.99 * XBOX, 0.99 * XBOX)

    def run():
        s = capi.Solver(12, 4, N, batch, structured=True)
        s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=p.x_min, xmax=p.x_max)
        s.set_reference(p.x_ref, p.u_ref)
        s.update_initialization(X0)
        s.calculate()
        r = s.get_results(want=("u", "status", "polish_iters"))
        s.close()
        return r
    on, off = _both("ALMPC_SDUAL_NO_SCREEN", run)
    assert np.array_equal(on["status"], off["status"])
    assert set(np.unique(on["status"])) <= {0, 3}
    ok = on["status"] == 0
    assert np.abs(on["u"][ok] - off["u"][ok]).max() <= 1e-9
    screened = np.flatnonzero((on["status"] == 3) & (on["polish_iters"] == 0))   # decided without a single working-set change
    assert len(screened) >= 100 and (on["status"] == 3).sum() > len(screened) * 0 + 100
    for i in list(screened[:6]) + list(screened[-6:]):
        assert mo.feasibility_slack(p, X0[i]) > 1e-9, i          # the phase-1 LP agrees: no admissible input sequence
    for i in np.flatnonzero(ok)[:4]:
        assert mo.feasibility_slack(p, X0[i]) <= 1e-9, i


def test_tier_handover_of_the_inverse_changes_nothing(capi, mo):
    N, batch = 50, 768
    p = mo.make_problem(*mo.quadrotor_model(), N, mo.quadrotor().u_min, mo.quadrotor().u_max, x_min=-XBOX, x_max=XBOX, terminal="equality")
    X0 = np.clip(_mixed_x0(mo, batch, first=4096), -0.99 * XBOX, 0.99 * XBOX)

    def run():
        os.environ["ALMPC_SDUAL_NO_SCREEN"] = "1"      # (so that the infeasible instances do walk the tiers)
        try:
            s = capi.Solver(12, 4, N, batch, structured=True)
            s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=p.x_min, xmax=p.x_max, terminal="equality")
            s.set_reference(p.x_ref, p.u_ref)
            s.update_initialization(X0)
            s.calculate()
            r = s.get_results(want=("u", "status", "polish_iters"))
            s.close()
        finally:
            os.environ.pop("ALMPC_SDUAL_NO_SCREEN", None)
        return r
    on, off = _both("ALMPC_SDUAL_NO_SINV_HANDOVER", run)
    assert np.array_equal(on["status"], off["status"]) and set(np.unique(on["status"])) <= {0, 3}
    assert (on["status"] == 3).sum() >= 20                # infeasible instances: their verdict needs ~70 rows, i.e. all three tiers
    ok = on["status"] == 0
    assert np.abs(on["u"][ok] - off["u"][ok]).max() <= 1e-8
    i = int(np.flatnonzero(ok & (on["polish_iters"] >= 10))[0])
    assert np.abs(on["u"][i] - mo.solve_mpc_exact(p, X0[i])["u"]).max() <= 1e-6


def test_redo_started_from_the_finishs_working_set_reaches_the_same_verdicts(capi, mo):
    p = mo.quadrotor()
    batch = 4096
    xmax = np.array([1, 1, 1, .5, .5, .5, .1, .1, .1, .1, .1, .1]) * 3.0
    X0 = np.clip(mo.quadrotor_x0_batch(batch, 1.0), -0.99 * xmax, 0.99 * xmax)

    def run():
        s = capi.Solver(12, 4, 30, batch)
        s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=-xmax, xmax=xmax, rho=30.0, rho_profile="stiffness", terminal="equality")
        s.set_reference(p.x_ref, p.u_ref)
        s.update_initialization(X0)
        s.calculate(capi.default_opts(rho=30.0, max_iter=8, check_every=8))
        r = s.get_results(want=("u", "status"))
        s.close()
        return r
    on, off = _both("ALMPC_NO_REDO_START", run)
    assert np.array_equal(on["status"], off["status"]) and set(np.unique(on["status"])) <= {0, 3}
    ok = on["status"] == 0
    assert np.abs(on["u"][ok] - off["u"][ok]).max() <= 1e-9


@pytest.mark.parametrize("box,cap", [(1.0, 0), (3.0, 3), (3.0, 12)])
def test_redo_start_built_in_registers_equals_the_bordered_one(capi, mo, box, cap):
    """k_sdual_start (the start's inverse as a Gauss-Jordan sweep in registers, from the cached responses) against k_sdual's own
    row-by-row bordering of the same list (ALMPC_SDUAL_NO_START_BUILD).  box 1.0: the benchmark's edge-of-feasibility instances (the redo
    ends in an infeasibility verdict); box 3.0 with the finish capped at `cap` changes: feasible instances handed over half-way, whose
    redo must end in the optimum -- held against the exact solver of the oracle as well."""
    p = mo.quadrotor()
    batch = 1024
    xmax = box * np.array([1, 1, 1, .5, .5, .5, .1, .1, .1, .1, .1, .1]) * 3.0
    X0 = np.clip(mo.quadrotor_x0_batch(batch, 1.0), -0.99 * xmax, 0.99 * xmax)

    def run(fallback=None):
        s = capi.Solver(12, 4, 30, batch, structured_fallback=fallback)
        s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=-xmax, xmax=xmax, rho=30.0, rho_profile="stiffness", terminal="equality")
        s.set_reference(p.x_ref, p.u_ref)
        s.update_initialization(X0)
        o = capi.default_opts(rho=30.0, max_iter=8, check_every=8)
        if cap:
            o.polish_max_iter = cap
        s.calculate(o)
        r = s.get_results(want=("u", "status", "polish_iters"))
        s.close()
        return r
    built, bordered = _both("ALMPC_SDUAL_NO_START_BUILD", run)
    assert np.array_equal(built["status"], bordered["status"]) and set(np.unique(built["status"])) <= {0, 3}
    ok = built["status"] == 0
    assert np.abs(built["u"][ok] - bordered["u"][ok]).max() <= 1e-9
    if cap:
        alone = run(fallback=False)   # what the finish alone leaves: the redo must have had work to do
        redone = np.flatnonzero((alone["status"] == 1) & ok)
        assert len(redone) >= (20 if cap <= 3 else 1)
        pb = mo.make_problem(*mo.quadrotor_model(), 30, p.u_min, p.u_max, x_min=-xmax, x_max=xmax, terminal="equality")
        for i in redone[:3]:
            assert np.abs(built["u"][i] - mo.solve_mpc_exact(pb, X0[i])["u"]).max() <= 1e-6


@pytest.mark.parametrize("shape", ["double_integrator", "random_6x3"])
def test_redo_start_on_other_shapes_than_the_benchmarks(capi, mo, shape):
    """k_sdual_start is written for any stage size (its coordinates come from SdualStartParams::SP / TP): the finish capped at 2
    changes on a double integrator (NT 2, MC 2 build) and on a random 6-state, 3-input plant (NT 6 -> the 8 x 4 build) with a
    state box and the terminal equality -- every instance handed over half-way; the redo, started from the finish's rows with the
    inverse built in registers, against the bordered start and against the exact oracle."""
    if shape == "double_integrator":
        A, B = np.array([[1.0, 1.0], [0.0, 1.0]]), np.array([[0.5], [1.0]])
        p = mo.make_problem(A, B, 20, [-1.0], [1.0], x_min=[-12.0, -1.5], x_max=[12.0, 1.5], terminal="equality")
        X0 = np.stack([np.linspace(-6.0, 6.0, 96), np.tile([0.0, 0.4, -0.4], 32)], axis=1)
    else:
        rng = np.random.default_rng(65)
        A = rng.standard_normal((6, 6)); A *= 0.95 / np.max(np.abs(np.linalg.eigvals(A)))
        B = rng.standard_normal((6, 3))
        p = mo.make_problem(A, B, 16, -0.3 * np.ones(3), 0.3 * np.ones(3), x_min=-4.0 * np.ones(6), x_max=4.0 * np.ones(6), terminal="equality", q=10.0, r=1.0)
        X0 = 0.8 * rng.standard_normal((96, 6))
    batch = len(X0)

    def run(cap, fallback=None):
        s = capi.Solver(p.n, p.m, p.N, batch, structured_fallback=fallback)
        s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=p.x_min, xmax=p.x_max, terminal="equality")
        s.set_reference(p.x_ref, p.u_ref)
        s.update_initialization(X0)
        o = capi.default_opts(max_iter=2, check_every=2)   # (a poor guess of the working set: the finish has changes to make)
        o.polish_max_iter = cap
        s.calculate(o)
        r = s.get_results(want=("u", "status"))
        s.close()
        return r
    built, bordered = _both("ALMPC_SDUAL_NO_START_BUILD", lambda: run(2))
    assert np.array_equal(built["status"], bordered["status"]) and set(np.unique(built["status"])) <= {0, 3}
    ok = built["status"] == 0
    assert ok.sum() >= batch // 3
    assert np.abs(built["u"][ok] - bordered["u"][ok]).max() <= 1e-9
    alone = run(2, fallback=False)
    redone = np.flatnonzero((alone["status"] == 1) & ok)
    assert len(redone) >= 8, np.bincount(alone["status"], minlength=4)
    for i in redone[:4]:
        assert np.abs(built["u"][i] - mo.solve_mpc_exact(p, X0[i])["u"]).max() <= 1e-6
    for i in np.flatnonzero(built["status"] == 3)[:3]:
        with pytest.raises(ValueError):
            mo.solve_mpc_exact(p, X0[i])


def test_instance_that_outgrows_the_finish_is_redone_from_the_finishs_rows(capi, mo):
    """Amplitude 3, state box 3 x the x0 scale, no terminal equality: instance 556 of the benchmark batch needs 65 rows -- one more than
    the state-row finish holds.  The finish gives up for want of room and (round 5) still hands over its rows; the start kernel builds
    their inverse for the LARGEST tier and the 64-row tier passes the start on as it stands: 2 changes instead of 60 from scratch
    (1.04 -> 0.26 ms).  Same answer as the redo from scratch (ALMPC_NO_REDO_START=1) and as the oracle's exact solver."""
    p = mo.quadrotor()
    batch = 640
    xmax = 3.0 * np.array([1, 1, 1, .5, .5, .5, .1, .1, .1, .1, .1, .1]) * 3.0
    X0 = np.clip(mo.quadrotor_x0_batch(batch, 3.0), -0.99 * xmax, 0.99 * xmax)

    def run():
        s = capi.Solver(12, 4, 30, batch)
        s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=-xmax, xmax=xmax, rho=30.0, rho_profile="stiffness")
        s.set_reference(p.x_ref, p.u_ref)
        s.update_initialization(X0)
        s.calculate(capi.default_opts(rho=30.0, max_iter=8, check_every=8))
        r = s.get_results(want=("u", "x", "status", "polish_iters"))
        s.close()
        return r
    start, scratch = _both("ALMPC_NO_REDO_START", run)
    assert np.array_equal(start["status"], scratch["status"]) and set(np.unique(start["status"])) <= {0, 3}
    ok = start["status"] == 0
    assert np.abs(start["u"][ok] - scratch["u"][ok]).max() <= 1e-8
    i = 556
    assert start["status"][i] == 0 and scratch["status"][i] == 0   # (polish_iters counts the LAST tier's changes only: not comparable)
    pb = mo.make_problem(*mo.quadrotor_model(), 30, p.u_min, p.u_max, x_min=-xmax, x_max=xmax)
    e = mo.solve_mpc_exact(pb, X0[i])   # (certified by the method-independent KKT test; a second opinion where its own inverse loses the instance)
    assert np.abs(start["u"][i] - e["u"]).max() <= 1e-6
    nrows = int(((e["u"] >= p.u_max[:, None] - 1e-9) | (e["u"] <= p.u_min[:, None] + 1e-9)).sum() +
                ((np.abs(e["x"][:, 1:]) >= xmax[:, None] - 1e-9)).sum())
    assert nrows >= 64   # (the capacity of the finish: one more row, weakly active, is what its 64-row build runs out of room with)


def test_redo_enqueued_ahead_of_the_look_equals_the_redo_after_it(capi, mo):
    """A synchronous look that found undecided instances makes the NEXT step's redo go on the stream gated behind the step, before the
    host waits (wait_and_settle; ALMPC_NO_PREDICTED_REDO=1: always after the look).  Four looked-at steps of the tight box + equality
    batch, the last two from a moved x0 that leaves nothing undecided: same results either way, never an undecided instance."""
    p = mo.quadrotor()
    batch = 2048
    xmax = np.array([1, 1, 1, .5, .5, .5, .1, .1, .1, .1, .1, .1]) * 3.0
    X0 = np.clip(mo.quadrotor_x0_batch(batch, 1.0), -0.99 * xmax, 0.99 * xmax)

    def run():
        s = capi.Solver(12, 4, 30, batch)
        s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=-xmax, xmax=xmax, rho=30.0, rho_profile="stiffness", terminal="equality")
        s.set_reference(p.x_ref, p.u_ref)
        o = capi.default_opts(rho=30.0, max_iter=8, check_every=8)
        out = []
        for x0 in (X0, X0, 0.25 * X0, 0.25 * X0, X0):
            s.update_initialization(x0)
            s.calculate(o)
            out.append(s.get_results(want=("u", "status")))
        s.close()
        return out
    ahead, after = _both("ALMPC_NO_PREDICTED_REDO", run)
    for a, b in zip(ahead, after):
        assert np.array_equal(a["status"], b["status"]) and set(np.unique(a["status"])) <= {0, 3}
        ok = a["status"] == 0
        assert np.abs(a["u"][ok] - b["u"][ok]).max() <= 1e-9
    assert (ahead[0]["status"] == 3).sum() > (ahead[2]["status"] == 3).sum()
    assert np.array_equal(ahead[0]["status"], ahead[4]["status"])


@pytest.mark.parametrize("m,N", [(4, 30), (3, 23)])      # nz 120 (packed triangle), nz 69 (odd: the full layout stays)
def test_packed_kkt_inverse_of_the_per_instance_step(capi, mo, m, N):
    n, batch = 6, 96
    rng = np.random.default_rng(5)
    A0 = 0.9 * np.eye(n) + 0.08 * rng.standard_normal((n, n))
    B0 = rng.standard_normal((n, m)) * 0.3
    A = np.repeat(A0[None], batch, 0) + 0.02 * rng.standard_normal((batch, n, n))
    B = np.repeat(B0[None], batch, 0) * (1.0 + 0.05 * rng.standard_normal((batch, 1, 1)))
    X0 = 2.0 * rng.standard_normal((batch, n))
    umin, umax = -0.4 * np.ones(m), 0.4 * np.ones(m)

    def run():
        s = capi.Solver(n, m, N, batch)
        s.design_batched(A, B, 100.0 * np.eye(n), 0.1 * np.eye(m), None, None, umin, umax)
        s.update_initialization(X0)
        s.calculate(capi.default_opts(max_iter=12, check_every=12))
        r = s.get_results(want=("u", "status", "iters"))
        s.calculate(capi.default_opts(max_iter=12, check_every=12, warm_start=1))
        w = s.get_results(want=("u", "status"))
        s.close()
        return r, w
    (on, onw), (off, offw) = _both("ALMPC_NO_PACKED_MINV", run)
    assert np.all(on["status"] == 0) and np.all(off["status"] == 0) and np.array_equal(on["iters"], off["iters"])
    assert np.abs(on["u"] - off["u"]).max() <= 1e-9 and np.abs(onw["u"] - offw["u"]).max() <= 1e-9
    for i in range(0, batch, 17):
        pi = mo.make_problem(A[i], B[i], N, umin, umax)
        assert np.abs(on["u"][i] - mo.solve_mpc_exact(pi, X0[i])["u"]).max() <= 1e-6, i
