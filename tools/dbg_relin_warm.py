#!/usr/bin/env python3
"""Debug aid: warm vs cold steps of the re-linearisation pipeline in closed loop; where they differ, which is right (oracle)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import almpc_loader
capi = almpc_loader.load_package()._capi
import mpc_oracle as mo
f = mo.synthetic_fnn()
batch, N, n, m = 256, 20, 4, 2
x_ref = np.array([0.2, -0.1, 0.05, 0.0])[:, None] * np.ones((n, N + 1))
u_ref = np.array([0.1, -0.2])[:, None] * np.ones((m, N))
Q, R = 100.0 * np.eye(n), 0.1 * np.eye(m)
Al, Bl = capi.fnn_linearize(f.W_in, f.W_h, f.b_h, f.W_out, x_ref[:, -1][None], u_ref[:, -1][None], act=f.act)
P = capi.dare(Al[0], Bl[0], Q, R)
X0 = x_ref[:, 0][None, :] + mo.splitmix_normal(0x5EED0004, 11, batch, n) * 1.5
fb = "--no-fallback" not in sys.argv
sw = capi.Solver(n, m, N, batch, timing=True, structured_fallback=fb)
sc = capi.Solver(n, m, N, batch, timing=True, structured_fallback=fb)
for s in (sw, sc):
    s.relin_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, None, P, [-1, -1], [1, 1], act=f.act)
    s.update_initialization(X0)
warm, cold = capi.default_opts(warm_start=1), capi.default_opts()
for step in range(8):
    sw.relin_fnn_step(warm)
    a = sw.get_results()
    x = a["x"][:, :, 0].copy()
    sc.update_initialization(x)
    sc.relin_fnn_step(cold)
    b = sc.get_results()
    d = np.abs(a["u"] - b["u"]).reshape(batch, -1).max(axis=1)
    bad = np.argsort(-d)[:4]
    print(f"step {step}: status warm {np.bincount(a['status'], minlength=4)} cold {np.bincount(b['status'], minlength=4)} max diff {d.max():.2e}"
          f"  polish its warm max {a['polish_iters'].max()} cold max {b['polish_iters'].max()}  timing warm {sw.relin_fnn_timing()} cold {sc.relin_fnn_timing()}")
    for i in bad:
        if d[i] < 1e-9: continue
        Ai, Bi = f.jacobian(x[i], u_ref[:, 0])
        p = mo.make_problem(Ai, Bi, N, [-1, -1], [1, 1], x_ref=x_ref, u_ref=u_ref, P=P)
        e = mo.solve_mpc_exact(p, x[i])
        print(f"   inst {i}: diff {d[i]:.2e} rho(A) {np.abs(np.linalg.eigvals(Ai)).max():.3f} warm-oracle {np.abs(a['u'][i]-e['u']).max():.2e} "
              f"cold-oracle {np.abs(b['u'][i]-e['u']).max():.2e} status {a['status'][i]} {b['status'][i]} pits {a['polish_iters'][i]} {b['polish_iters'][i]}")
    sw.relin_fnn_advance()
