"""Time the state-row path (k_polish_gen): quadrotor with a state box, batch 4096: python tools/time_state_rows.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo, bench
capi = almpc_loader.load_package()._capi
p = mo.quadrotor(); b = 4096
CASES = ((1.0, 3.0), (1.0, 1.0), (3.0, 3.0))
if len(sys.argv) > 1:  # one case only (for rocprofv3): python tools/time_state_rows.py <case 0..2>
    CASES = (CASES[int(sys.argv[1])],)
for amp, box in CASES:
    X0 = mo.quadrotor_x0_batch(b, amp)
    xmax = box * np.array([1, 1, 1, .5, .5, .5, .1, .1, .1, .1, .1, .1]) * 3.0
    X0 = np.clip(X0, -0.99 * xmax, 0.99 * xmax)
    for terminal in ("none", "equality"):
        fb = {"0": False, "1": True}.get(os.environ.get("FALLBACK", ""), None)   # FALLBACK=0: what the condensed finish alone leaves
        s = capi.Solver(12, 4, 30, b, timing=True, structured_fallback=fb)
        s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=-xmax, xmax=xmax, rho=30.0, rho_profile="stiffness", terminal=terminal)
        s.set_reference(p.x_ref, p.u_ref); s.update_initialization(X0)
        o = capi.default_opts(rho=30.0, max_iter=8, check_every=8)
        for _ in range(3): s.calculate(o)
        s.timing_reset(20)
        t0 = time.perf_counter()
        for _ in range(20): s.calculate(o, sync=False)
        s.synchronize()
        el = (time.perf_counter() - t0) / 20
        ts = s.timing_summary()
        r = s.get_results(want=("status", "polish_iters", "x"))
        print(f"amp {amp} box x{box} terminal {terminal}: step {1e6*el:.0f} us (admm {1e3*ts['admm_ms']/ts['steps']:.0f}, finish {1e3*ts['polish_ms']/ts['steps']:.0f}), status {np.bincount(r['status'], minlength=4).tolist()}, "
              f"finish its mean {r['polish_iters'].mean():.1f} max {r['polish_iters'].max()}")
        if fb is False and (r["status"] == 1).any():
            bad = np.flatnonzero(r["status"] == 1)
            print(f"    left unsolved by the finish: {len(bad)} instances, finish iterations there {sorted(r['polish_iters'][bad].tolist())[-12:]}")
        s.close()
