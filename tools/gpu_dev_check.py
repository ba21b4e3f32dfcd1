"""Developer check on a GPU box: HIP path vs oracle on configs 1, QTP, quadrotor; prints diagnostics."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import almpc_loader
import mpc_oracle as mo, c_oracle as co
pkg = almpc_loader.load_package(); capi = pkg._capi

def run(p, X0, label, timing=True, **optkw):
    X0 = np.atleast_2d(X0); b = X0.shape[0]
    s = capi.Solver(p.n, p.m, p.N, b, timing=timing)
    t = time.time(); s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max); td = time.time() - t
    s.set_reference(p.x_ref, p.u_ref)
    des = s.get_design(); od = mo.design_shared(p)
    print(f"[{label}] design {td*1e3:.1f} ms  |H-Ho|/|Ho| {np.abs(des['H']-od['H']).max()/np.abs(od['H']).max():.2e}  |F-Fo|/|Fo| {np.abs(des['F']-od['F']).max()/max(1e-300,np.abs(od['F']).max()):.2e}  |P-Po| {np.abs(des['P']-p.P).max():.2e} |d-do|/do {np.abs(des['d']/od['d']-1).max():.2e}")
    s.update_initialization(X0)
    opts = capi.default_opts(**optkw)
    s.calculate(opts); s.calculate(opts)
    r = s.get_results(); tm = s.get_timing()
    ref = co.step_batch(p, od, X0, max_iter=opts.max_iter, check_every=opts.check_every, polish=bool(opts.polish))
    nex = min(b, 64)
    ex = [mo.solve_mpc_exact(p, X0[i]) for i in range(nex)]
    eu = max(np.abs(r['u'][i] - ex[i]['u']).max() for i in range(nex)); exx = max(np.abs(r['x'][i] - ex[i]['x']).max() for i in range(nex))
    print(f"[{label}] batch {b}: max|u-u*| {eu:.2e} max|x-x*| {exx:.2e} (first {nex}) | vs C oracle u {np.abs(r['u']-ref['u']).max():.2e} x {np.abs(r['x']-ref['x']).max():.2e} e_x {np.abs(r['e_x']-ref['e_x']).max():.2e} e_u {np.abs(r['e_u']-ref['e_u']).max():.2e}")
    print(f"[{label}] status {np.bincount(r['status'], minlength=3)} iters {np.unique(r['iters'], return_counts=True)} iters-mismatch {(r['iters']!=ref['iters']).sum()} polish mean {r['polish_iters'].mean():.2f} max {r['polish_iters'].max()} pit-mismatch {(r['polish_iters']!=ref['polish_iters']).sum()}")
    print(f"[{label}] timing {tm}")
    s.close()
    return r

p1 = mo.double_integrator()
run(p1, np.array([[1., 0.]]), "cfg1 x0=(1,0)")
run(p1, np.array([[5., 0.]]), "cfg1 x0=(5,0)")
blob = open(os.path.join(ROOT, "tests/golden/linear_regressor_train_result.jls"), "rb").read()
A, B = mo.decode_linear_regressor_fixture(blob)
run(mo.qtp_linear_fixture_problem(A, B), np.full((1, 4), 0.6), "qtp")
Q = mo.quadrotor()
for s_ in (0.3, 1.0, 3.0, 10.0):
    run(Q, mo.quadrotor_x0_batch(100, s_), f"quad s={s_} b=100")
for s_ in (1.0, 3.0):
    run(Q, mo.quadrotor_x0_batch(4096, s_), f"quad s={s_} b=4096")
run(Q, mo.quadrotor_x0_batch(4096, 1.0), "quad s=1 b=4096 K=25", max_iter=25)
