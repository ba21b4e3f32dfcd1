"""Headline workload (4096 quadrotor instances, mixed amplitudes, cold start every step): batch-steps/s over (ADMM iterations K, rho of the
stiffness profile).  python tools/sweep_operating_point.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, almpc_loader, bench
pkg = almpc_loader.load_package(); capi = pkg._capi
import importlib
wl = importlib.import_module(pkg.__name__ + ".workloads")
p = wl.quadrotor(30); B = 4096
X0 = bench.make_x0(wl, 0, B)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for rho in (30.0, 45.0, 60.0, 80.0):
    s = capi.Solver(12, 4, 30, B)
    s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, rho=rho, rho_profile="stiffness")
    s.set_reference(p.x_ref, p.u_ref); s.update_initialization(X0)
    row = []
    for K in (5, 6, 7, 8):
        o = capi.default_opts(rho=rho, max_iter=K, check_every=K, keep_warm_state=False)
        for _ in range(20): s.calculate(o, sync=False)
        s.synchronize()
        best = 1e9
        for _rep in range(3):
            t0 = time.perf_counter()
            for _ in range(steps): s.calculate(o, sync=False)
            s.synchronize()
            best = min(best, time.perf_counter() - t0)
        r = s.get_results(want=("status", "polish_iters"))
        row.append(f"K{K}: {steps / best / 1e3:6.2f}k (unsolved {int((r['status'] != 0).sum())}, its max {int(r['polish_iters'].max())})")
    print(f"rho {rho:5.1f}  " + "  ".join(row), flush=True)
    s.close()
