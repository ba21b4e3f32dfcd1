"""GPU tests of the group forms of everything a handle can do (include/almpc.h almpc_group_*; VERDICT round 3, missing #3): the reference
API is one process, one call (proceed_controller, src/main/main_mpc.jl:22-53), so a one-process multi-GPU caller must not have to
drive the handles by hand.  One GPU here: a group of two (three) handles on device 0 must give what ONE handle gives on the
concatenated batch -- per-instance designs, the re-linearisation pipeline (BASELINE configs[3]), the SQP loop (configs[4]), state rows,
structured handles, asynchronous tickets and zero-copy x0 slots."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fnn_problem(mo, batch, N, act="relu", seed=0x5EED0004):
    f = mo.synthetic_fnn(act=act)
    n, m = 4, 2
    x_ref = np.array([0.2, -0.1, 0.05, 0.0])[:, None] * np.ones((n, N + 1))
    u_ref = np.array([0.1, -0.2])[:, None] * np.ones((m, N))
    X0 = x_ref[:, 0][None, :] + 0.6 * mo.splitmix_normal(seed, 0, batch, n)
    return f, n, m, x_ref, u_ref, X0


def test_group_design_batched_with_state_box(capi, mo):
    rng = np.random.default_rng(11)
    b, n, m, N = 37, 3, 2, 12
    As = np.stack([(lambda A: A * (0.9 / np.max(np.abs(np.linalg.eigvals(A)))))(rng.standard_normal((n, n))) for _ in range(b)])
    Bs = rng.standard_normal((b, n, m))
    X0 = rng.standard_normal((b, n))
    P = np.stack([mo.dare(As[i], Bs[i], 100.0 * np.eye(n), 0.1 * np.eye(m)) for i in range(b)])
    kw = dict(xmin=-4.0 * np.ones(n), xmax=4.0 * np.ones(n))
    one = capi.Solver(n, m, N, b)
    one.design_batched(As, Bs, 100.0 * np.eye(n), 0.1 * np.eye(m), None, P, -np.ones(m), np.ones(m), **kw)
    one.update_initialization(X0); one.calculate()
    ref = one.get_results()
    one.close()
    g = capi.Group(n, m, N, b, devices=[0, 0, 0])
    g.design_batched(As, Bs, 100.0 * np.eye(n), 0.1 * np.eye(m), None, P, -np.ones(m), np.ones(m), **kw)
    g.update_initialization(X0)
    g.calculate()
    got = g.get_results()
    g.close()
    for k in ("status", "u", "x", "polish_iters"):
        assert np.array_equal(got[k], ref[k]), k
    assert set(np.unique(ref["status"])) <= {0, 3}
    assert (np.abs(ref["x"][ref["status"] == 0]) <= 4.0 + 1e-7).all()


def test_group_relin_pipeline_and_closed_loop(capi, mo):
    batch, N = 203, 20
    f, n, m, x_ref, u_ref, X0 = _fnn_problem(mo, batch, N)
    Q, R = 100.0 * np.eye(n), 0.1 * np.eye(m)
    Al, Bl = capi.fnn_linearize(f.W_in, f.W_h, f.b_h, f.W_out, x_ref[:, -1][None], u_ref[:, -1][None], act=f.act)
    P = capi.dare(Al[0], Bl[0], Q, R)
    one = capi.Solver(n, m, N, batch)
    one.relin_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, None, P, [-1, -1], [1, 1], act=f.act)
    g = capi.Group(n, m, N, batch, devices=[0, 0])
    g.relin_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, None, P, [-1, -1], [1, 1], act=f.act)
    one.update_initialization(X0); g.update_initialization(X0, resident=True)
    warm = capi.default_opts(warm_start=1)
    for step in range(3):   # cold step, then two closed-loop steps on the network itself with warm starts
        o = None if step == 0 else warm
        one.relin_fnn_step(o); g.relin_fnn_step(o, sync=(step != 1))
        if step == 1:
            g.synchronize()
        a, b_ = one.get_results(), g.get_results()
        for k in ("status", "u", "x", "iters", "polish_iters"):
            assert np.array_equal(a[k], b_[k]), (step, k)
        assert np.all(a["status"] == 0)
        one.relin_fnn_advance(); g.relin_fnn_advance()
    one.close(); g.close()


def test_group_sqp_loop(capi, mo):
    batch, N, iters = 21, 20, 8
    f, n, m, x_ref, u_ref, X0 = _fnn_problem(mo, batch, N, act="tanh", seed=0x5EED0005)
    Q, R, S = 100.0 * np.eye(n), 0.1 * np.eye(m), 0.2 * np.eye(m)
    P = np.repeat((150.0 * np.eye(n))[None], batch, axis=0) * (1.0 + 0.01 * np.arange(batch))[:, None, None]   # per-instance terminal weights
    ug = 0.3 * mo.splitmix_normal(0x5EED0008, 0, batch, m * N).reshape(batch, m, N)
    for qp in ("condensed", "structured"):
        one = capi.Solver(n, m, N, batch)
        one.sqp_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, S, P, -np.ones(m), np.ones(m), act="tanh", qp_solver=qp)
        one.sqp_fnn_start(X0, ug)
        st1, de1 = one.sqp_fnn_iterate(iters, step_rule="merit")
        a = one.get_results()
        one.close()
        g = capi.Group(n, m, N, batch, devices=[0, 0])
        g.sqp_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, S, P, -np.ones(m), np.ones(m), act="tanh", qp_solver=qp)
        g.sqp_fnn_start(X0, ug)
        st2, de2 = g.sqp_fnn_iterate(iters, step_rule="merit")
        b_ = g.get_results()
        assert g.sqp_fnn_skipped().sum() == 0
        g.close()
        np.testing.assert_array_equal(st1, st2); np.testing.assert_array_equal(de1, de2)
        for k in ("u", "x", "status"):
            assert np.array_equal(a[k], b_[k]), (qp, k)


def test_group_of_structured_handles_tickets_and_staged_x0(capi, mo):
    q = mo.quadrotor(50)
    xmax = np.array([3, 3, 3, 1.5, 1.5, 1.5, 0.3, 0.3, 0.3, 1.0, 1.0, 1.0])
    p = mo.make_problem(q.A, q.B, 50, q.u_min, q.u_max, x_min=-xmax, x_max=xmax, s=2.0)
    batch = 150
    X0 = np.clip(mo.quadrotor_x0_batch(batch, 1.0), -0.99 * xmax, 0.99 * xmax)
    one = capi.Solver(p.n, p.m, p.N, batch, structured=True)
    one.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=p.x_min, xmax=p.x_max)
    one.update_initialization(X0); one.calculate()
    ref = one.get_results()
    one.close()
    g = capi.Group(p.n, p.m, p.N, batch, devices=[0, 0], structured=True)
    g.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=p.x_min, xmax=p.x_max)
    g.set_reference(p.x_ref, p.u_ref)
    slots = g.x0_staging()
    for (f0, c), sl in zip(g.shards, slots):
        sl[...] = X0[f0:f0 + c]
    g.update_initialization_staged()
    g.calculate(sync=False)
    t = g.get_results_async(want=("u0", "status", "u"))
    got = g.get_results_wait(t, want=("u0", "status", "u"))
    g.close()
    assert np.array_equal(got["status"], ref["status"]) and np.array_equal(got["u"], ref["u"])
    assert np.array_equal(got["u0"], ref["u"][:, :, 0])
    assert set(np.unique(ref["status"])) <= {0, 3} and (ref["status"] == 0).sum() >= 100
