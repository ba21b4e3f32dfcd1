"""Wall-clock steps/s with and without the per-step HIP events (ALMPC_FLAG_TIMING)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, bench, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
p = mo.quadrotor(); X0 = bench.make_x0(mo, 0, 4096)
for timing in (False, True):
    s = capi.Solver(12, 4, 30, 4096, timing=timing)
    s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, rho=30.0, rho_profile="stiffness"); s.set_reference(p.x_ref, p.u_ref); s.update_initialization(X0)
    o = capi.default_opts(rho=30.0, max_iter=8, check_every=8)
    if timing: s.timing_reset(700)
    for _ in range(50): s.calculate(o, sync=False)
    s.synchronize(); t = time.perf_counter()
    for _ in range(500): s.calculate(o, sync=False)
    s.synchronize(); dt = time.perf_counter() - t
    print("timing events", timing, ": %.1f us/step -> %.0f steps/s" % (dt / 500 * 1e6, 500 / dt))
    s.close()
