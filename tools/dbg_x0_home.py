import os, sys, time
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, bench, importlib
pkg = almpc_loader.load_package(); capi = pkg._capi
wl = importlib.import_module(pkg.__name__ + ".workloads")
p = wl.quadrotor(30); B = 4096
s = capi.Solver(12, 4, 30, B); s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, rho=45.0, rho_profile="stiffness")
s.set_reference(p.x_ref, p.u_ref)
opts = capi.default_opts(rho=45.0, max_iter=6, check_every=6, keep_warm_state=False)
X0 = bench.make_x0(wl, 0, B, None)
for mode in ("device", "zero-copy", "device", "zero-copy"):
    if mode == "device": s.update_initialization(X0)
    else: s.update_initialization_async(X0)
    for _ in range(300): s.calculate(opts, sync=False)
    s.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(200): s.calculate(opts, sync=False)
        s.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(mode, "%.2f us/step" % (best / 200 * 1e6))
