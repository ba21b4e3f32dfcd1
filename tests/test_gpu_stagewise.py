"""GPU tests of the stage-wise dual active-set solve, k_sdual (csrc/almpc_sdual.hip.h): the multiple-shooting form the reference
builds -- dynamics as constraints, input box, state box on every stage (..linear.jl:48-78), terminal equality
(src/sub/design_mpc.jl:330-331), input-rate weight S (src/sub/design_mpc.jl:423-446) -- on structured handles, i.e. also beyond the
condensed kernels' m N <= 128 (quadrotor at N = 50: m N = 200).
Oracles: mpc_oracle.solve_mpc_exact (the condensed exact solver with its method-independent KKT certificate; raises ValueError for an
infeasible problem) at 1e-6 on u, and the numpy restatement of the same algorithm (stagewise_oracle.solve_stage_dual: same decisions,
so the iteration counts agree)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

U_TOL = 1e-6

XMAX = np.array([3, 3, 3, 1.5, 1.5, 1.5, 0.3, 0.3, 0.3, 1.0, 1.0, 1.0])


@pytest.fixture(scope="module")
def so():
    import stagewise_oracle
    return stagewise_oracle


def _solve(capi, p, X0, guess_from=None, **kw):
    s = capi.Solver(p.n, p.m, p.N, len(X0), structured=True)
    s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=p.x_min, xmax=p.x_max, terminal=p.terminal)
    s.set_reference(p.x_ref, p.u_ref)
    s.update_initialization(X0)
    s.calculate()
    r = s.get_results()
    s.close()
    return r


def _check_against_exact(mo, p, X0, r, sample):
    """status and u of the sampled instances against the exact oracle: 0 <-> solved within U_TOL, 3 <-> the oracle raises ValueError"""
    n_inf = 0
    for i in sample:
        try:
            e = mo.solve_mpc_exact(p, X0[i])
        except ValueError:
            assert r["status"][i] == 3, (i, r["status"][i])
            n_inf += 1
            continue
        assert r["status"][i] == 0, (i, r["status"][i])
        assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL, (i, np.abs(r["u"][i] - e["u"]).max())
        assert np.abs(r["x"][i] - e["x"]).max() <= 1e-5
    return n_inf


def _properties(p, X0, r):
    """every solved instance: dynamics, input box, state box, terminal equality"""
    ok = r["status"] == 0
    ex, eu = r["e_x"][ok], r["e_u"][ok]
    pred = np.einsum("ij,bjk->bik", p.A, ex[:, :, :-1]) + np.einsum("ij,bjk->bik", p.B, eu)
    assert np.abs(pred - ex[:, :, 1:]).max() <= 1e-8 * max(1.0, np.abs(ex).max())
    assert np.all(r["u"][ok] >= p.u_min[None, :, None]) and np.all(r["u"][ok] <= p.u_max[None, :, None])
    if p.x_min is not None:
        assert np.all(r["x"][ok] >= p.x_min[None, :, None] - 1e-7) and np.all(r["x"][ok] <= p.x_max[None, :, None] + 1e-7)
    if p.terminal == "equality":
        assert np.abs(ex[:, :, -1]).max() <= 1e-7
    np.testing.assert_array_equal(r["x"][:, :, 0], X0)


@pytest.mark.parametrize("case", ["box", "box_only_inputs", "eq", "box_eq", "S", "S_box_tvref"])
def test_quadrotor_N30_against_the_exact_oracle(capi, mo, so, case):
    q = mo.quadrotor(30)
    kw = {}
    amp = 3.0
    if case in ("box", "box_eq", "S_box_tvref"):
        kw.update(x_min=-XMAX, x_max=XMAX)
    if case in ("eq", "box_eq"):
        kw.update(terminal="equality"); amp = 1.0
    if case in ("S", "S_box_tvref"):
        kw.update(s=5.0)
    if case == "S_box_tvref":
        kw.update(u_ref=0.01 * np.sin(np.arange(30))[None, :] * np.ones((4, 1)))
    p = mo.make_problem(q.A, q.B, 30, q.u_min, q.u_max, **kw)
    X0 = mo.quadrotor_x0_batch(96, amp)
    if p.x_min is not None:
        X0 = np.clip(X0, -0.99 * XMAX, 0.99 * XMAX)
    r = _solve(capi, p, X0)
    assert set(np.unique(r["status"])) <= {0, 3}
    _properties(p, X0, r)
    n_inf = _check_against_exact(mo, p, X0, r, range(0, 96, 4))
    if case in ("box_only_inputs", "eq", "S"):
        assert n_inf == 0 and np.all(r["status"] == 0)
    # same decisions as the numpy restatement
    for i in (0, 17, 50, 95):
        o = so.solve_mpc_stagewise(p, X0[i])
        assert o["status"] == r["status"][i]
        if o["status"] == 0:
            assert o["iters"] == r["polish_iters"][i], (i, o["iters"], r["polish_iters"][i])
            assert np.abs(r["u"][i] - o["u"]).max() <= 1e-9


@pytest.mark.parametrize("case", ["box", "eq", "S"])
def test_quadrotor_N50_state_rows_beyond_the_condensed_horizon(capi, mo, so, case):
    """m N = 200: no condensed handle exists for this shape (missing #1)."""
    q = mo.quadrotor(50)
    kw = dict(x_min=-XMAX, x_max=XMAX) if case == "box" else (dict(terminal="equality") if case == "eq" else dict(s=5.0))
    p = mo.make_problem(q.A, q.B, 50, q.u_min, q.u_max, **kw)
    X0 = np.concatenate([mo.quadrotor_x0_batch(48, a, first_instance=48 * k) for k, a in enumerate((0.3, 1.0, 3.0))])
    if case == "eq":
        X0 = X0[:96]
    if p.x_min is not None:
        X0 = np.clip(X0, -0.99 * XMAX, 0.99 * XMAX)
    r = _solve(capi, p, X0)
    assert set(np.unique(r["status"])) <= {0, 3}
    _properties(p, X0, r)
    n_inf = _check_against_exact(mo, p, X0, r, range(0, len(X0), 6))
    if case != "box":
        assert n_inf == 0 and np.all(r["status"] == 0)
    for i in (0, 60, len(X0) - 1):
        o = so.solve_mpc_stagewise(p, X0[i])
        assert o["status"] == r["status"][i]
        if o["status"] == 0:
            assert o["iters"] == r["polish_iters"][i]


def test_double_integrator_with_everything(capi, mo, so):
    """n = 2, m = 1 (padded to the (4, 2) build with S): box, equality and S together, infeasible instances named."""
    d = mo.double_integrator(10)
    p = mo.make_problem(d.A, d.B, 6, d.u_min, d.u_max, s=2.0, x_min=[-6.0, -1.5], x_max=[6.0, 1.5], terminal="equality")
    rng = np.random.default_rng(3)
    X0 = np.stack([rng.uniform(-5.9, 5.9, 64), rng.uniform(-1.4, 1.4, 64)], axis=1)
    r = _solve(capi, p, X0)
    assert set(np.unique(r["status"])) <= {0, 3}
    assert (r["status"] == 0).sum() >= 8 and (r["status"] == 3).sum() >= 2
    _properties(p, X0, r)
    _check_against_exact(mo, p, X0, r, range(64))


def test_working_sets_beyond_the_first_tier(capi, mo, so):
    """Heavily saturated instances: more than 32 rows in the working set (second launch with room for 64)."""
    q = mo.quadrotor(50)
    p = mo.make_problem(q.A, q.B, 50, q.u_min, q.u_max)
    X0 = mo.quadrotor_x0_batch(48, 6.0)
    r = _solve(capi, p, X0)
    assert np.all(r["status"] == 0)
    big = 0
    for i in range(0, 48, 4):
        e = mo.solve_mpc_exact(p, X0[i])
        assert np.abs(r["u"][i] - e["u"]).max() <= U_TOL
        v = e["e_u"]
        nact = int(((e["u"] >= p.u_max[:, None] - 1e-12) | (e["u"] <= p.u_min[:, None] + 1e-12)).sum())
        big += nact > 32
    assert big >= 1


def test_default_redo_on_the_shared_condensed_path_is_lazy_and_complete(capi, mo):
    """Condensed handle, shared model, input box only (the headline path).  A finish capped at two working-set changes leaves the
    saturated instances with ALMPC_MAX_ITER; by default (almpc_set_structured_fallback: auto) the stage-wise solvers redo them when
    the host looks at the results -- no launch on the step path --, switched off they stay as the finish left them."""
    p = mo.quadrotor(30)
    X0 = mo.quadrotor_x0_batch(256, 3.0)
    opts = capi.default_opts(polish_max_iter=2)
    out = {}
    for fb in (False, None, True):
        s = capi.Solver(12, 4, 30, len(X0), structured_fallback=fb)
        s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max)
        s.update_initialization(X0)
        s.calculate(opts)
        out[fb] = s.get_results()
        if fb is None:   # asynchronous step + synchronous getter: settled there
            s.update_initialization(0.9 * X0)
            s.calculate(opts, sync=False)
            u0 = s.get_first_input()
            r2 = s.get_results()
            assert np.all(r2["status"] == 0) and np.array_equal(u0, r2["u"][:, :, 0])
        s.close()
    assert (out[False]["status"] == 1).sum() >= 50
    for fb in (None, True):
        assert np.all(out[fb]["status"] == 0)
    assert np.array_equal(out[None]["u"], out[True]["u"])
    done = out[False]["status"] == 0
    assert np.array_equal(out[False]["u"][done], out[None]["u"][done])     # solved instances are not touched
    for i in np.flatnonzero(~done)[:12]:
        assert np.abs(out[None]["u"][i] - mo.solve_mpc_exact(p, X0[i])["u"]).max() <= U_TOL


def test_default_redo_with_state_rows_is_lazy_complete_and_uses_the_steps_own_x0(capi, mo):
    """Condensed handle with a tight state box and the terminal equality: the finish (k_polish_gen) leaves a few edge-of-feasibility
    instances without a verdict; by default the stage-wise redo decides them when the host looks at the results (status 0 or 3, never
    1), with the same outcome as the eager form -- also when the caller has handed over the NEXT x0 in between (the redo reads x0 from
    the step's own x)."""
    p = mo.quadrotor(30)
    xmax = np.array([1, 1, 1, .5, .5, .5, .1, .1, .1, .1, .1, .1]) * 3.0
    X0 = np.clip(mo.quadrotor_x0_batch(1024, 1.0), -0.99 * xmax, 0.99 * xmax)
    o = capi.default_opts(rho=30.0, max_iter=8, check_every=8)
    out = {}
    for fb in (False, None, True):
        s = capi.Solver(12, 4, 30, len(X0), structured_fallback=fb)
        s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=-xmax, xmax=xmax, rho=30.0, rho_profile="stiffness",
                        terminal="equality")
        s.set_reference(p.x_ref, p.u_ref)
        s.update_initialization(X0)
        s.calculate(o)
        out[fb] = s.get_results()
        if fb is None:
            s.calculate(o, sync=False)
            s.update_initialization(0.5 * X0)      # the next x0 arrives before anybody has looked at this step's results
            r2 = s.get_results()
            assert np.array_equal(r2["status"], out[None]["status"]) and np.array_equal(r2["u"], out[None]["u"])
            np.testing.assert_array_equal(r2["x"][:, :, 0], X0)
        s.close()
    left = out[False]["status"] == 1
    assert left.sum() >= 2
    for fb in (None, True):
        assert set(np.unique(out[fb]["status"])) <= {0, 3}
    assert np.array_equal(out[None]["status"], out[True]["status"]) and np.array_equal(out[None]["u"], out[True]["u"])
    decided = ~left
    assert np.array_equal(out[False]["status"][decided], out[None]["status"][decided])
    assert np.array_equal(out[False]["u"][decided], out[None]["u"][decided])
    q = mo.make_problem(p.A, p.B, 30, p.u_min, p.u_max, x_min=-xmax, x_max=xmax, terminal="equality")
    for i in np.flatnonzero(left)[:4]:
        try:
            e = mo.solve_mpc_exact(q, X0[i])
            assert out[None]["status"][i] == 0 and np.abs(out[None]["u"][i] - e["u"]).max() <= U_TOL
        except ValueError:
            assert out[None]["status"][i] == 3


def test_cached_responses_equal_the_sweeps(capi, mo, monkeypatch):
    """Shared model: the responses of the unconstrained problem are cached once per design (SdualParams::ghat) and a working-set change
    streams columns instead of running two sweeps.  Same decisions (statuses, change counts) and the same optimum as the sweep-only
    kernel (ALMPC_SDUAL_NO_GHAT), with a state box, the terminal equality and S."""
    q = mo.quadrotor(30)
    for kw, amp in ((dict(x_min=-XMAX, x_max=XMAX), 3.0), (dict(terminal="equality"), 1.0), (dict(s=5.0), 3.0)):
        p = mo.make_problem(q.A, q.B, 30, q.u_min, q.u_max, **kw)
        X0 = mo.quadrotor_x0_batch(128, amp)
        if p.x_min is not None:
            X0 = np.clip(X0, -0.99 * XMAX, 0.99 * XMAX)
        res = {}
        for tag in ("cached", "sweeps"):
            if tag == "sweeps":
                monkeypatch.setenv("ALMPC_SDUAL_NO_GHAT", "1")
            else:
                monkeypatch.delenv("ALMPC_SDUAL_NO_GHAT", raising=False)
            res[tag] = _solve(capi, p, X0)
        monkeypatch.delenv("ALMPC_SDUAL_NO_GHAT", raising=False)
        a, b = res["cached"], res["sweeps"]
        assert np.array_equal(a["status"], b["status"]) and set(np.unique(a["status"])) <= {0, 3}
        ok = a["status"] == 0
        assert ok.sum() >= 24
        assert np.array_equal(a["polish_iters"][ok], b["polish_iters"][ok])
        assert np.abs(a["u"][ok] - b["u"][ok]).max() <= 1e-9


def test_s0_table_of_the_state_row_finish_equals_the_rollout(capi, mo, monkeypatch):
    """Condensed handle with state rows, shared model and references: the state rows' s0 comes from the design-time affine map
    (k_s0_basis) instead of the prologue's rollout of v0 (ALMPC_NO_S0_BASIS): same statuses, same optimum -- also after a change of
    the references (the constant part of the map is rebuilt by almpc_set_reference)."""
    p = mo.quadrotor(30)
    xmax = np.array([1, 1, 1, .5, .5, .5, .1, .1, .1, .1, .1, .1]) * 3.0
    X0 = np.clip(mo.quadrotor_x0_batch(256, 1.0), -0.99 * xmax, 0.99 * xmax)
    u_ref = 0.01 * np.sin(np.arange(30))[None, :] * np.ones((4, 1))
    res = {}
    for tag in ("table", "rollout"):
        if tag == "rollout":
            monkeypatch.setenv("ALMPC_NO_S0_BASIS", "1")
        else:
            monkeypatch.delenv("ALMPC_NO_S0_BASIS", raising=False)
        s = capi.Solver(12, 4, 30, len(X0))
        s.design_shared(p.A, p.B, p.Q, p.R, 5.0 * np.eye(4), None, p.u_min, p.u_max, xmin=-xmax, xmax=xmax, rho=30.0, rho_profile="stiffness")
        out = []
        for ur in (np.zeros((4, 30)), u_ref):
            s.set_reference(p.x_ref, ur)
            s.update_initialization(X0)
            s.calculate(capi.default_opts(rho=30.0, max_iter=8, check_every=8))
            out.append(s.get_results())
        res[tag] = out
        s.close()
    monkeypatch.delenv("ALMPC_NO_S0_BASIS", raising=False)
    for a, b in zip(res["table"], res["rollout"]):
        assert np.array_equal(a["status"], b["status"]) and set(np.unique(a["status"])) <= {0, 3}
        ok = a["status"] == 0
        assert ok.sum() >= 64
        assert np.abs(a["u"][ok] - b["u"][ok]).max() <= 1e-9
