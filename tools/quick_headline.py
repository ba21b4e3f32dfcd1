"""Headline step rate of one library build (ALMPC_LIB), mixed batch and the three amplitude classes: for A/B runs of kernel variants."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, bench, importlib
pkg = almpc_loader.load_package(); capi = pkg._capi
wl = importlib.import_module(pkg.__name__ + ".workloads")
p = wl.quadrotor(30); B = 4096
FB = os.environ.get("FALLBACK")
s = capi.Solver(12, 4, 30, B, structured_fallback=None if FB is None else bool(int(FB))); s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, rho=45.0, rho_profile="stiffness")
s.set_reference(p.x_ref, p.u_ref)
opts = capi.default_opts(rho=45.0, max_iter=int(os.environ.get("K", "6")), check_every=int(os.environ.get("K", "6")), keep_warm_state=False)
out = []
for amp in (None, 0.3, 1.0, 3.0):
    s.update_initialization(bench.make_x0(wl, 0, B, amp))
    for _ in range(300): s.calculate(opts, sync=False)
    s.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(200): s.calculate(opts, sync=False)
        s.synchronize()
        best = min(best, time.perf_counter() - t0)
    st = s.get_results(want=("status",))["status"]
    out.append("%s: %.0f steps/s (%.2f us)%s" % ("mix" if amp is None else amp, 200 / best, best / 200 * 1e6, "" if (st == 0).all() else " UNSOLVED %d" % (st != 0).sum()))
print(os.path.basename(os.environ.get("ALMPC_LIB", "libalmpc.so")), "fallback", FB, " | ".join(out))
