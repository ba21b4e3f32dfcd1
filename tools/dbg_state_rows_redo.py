"""Debug aid: which instances of the state-row cases (tools/time_state_rows.py) the condensed finish leaves unsolved, what the
stage-wise redo makes of them and in how many working-set changes: python tools/dbg_state_rows_redo.py [case 0..5]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
if os.environ.get("STAMPS"):   # per-phase cycles of the redo (k_sdual) from the -DALMPC_STAMPS build
    os.environ["ALMPC_LIB"] = os.path.join(ROOT, "automationlabsmodelpredictivecontrol.jl_amd", "lib", "libalmpc_stamps.so")
import ctypes
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
p = mo.quadrotor(); b = 4096
cases = range(6) if len(sys.argv) < 2 else (int(sys.argv[1]),)
for case in cases:
    amp, box = ((1.0, 3.0), (1.0, 1.0), (3.0, 3.0))[case // 2]
    terminal = ("none", "equality")[case % 2]
    X0 = mo.quadrotor_x0_batch(b, amp)
    xmax = box * np.array([1, 1, 1, .5, .5, .5, .1, .1, .1, .1, .1, .1]) * 3.0
    X0 = np.clip(X0, -0.99 * xmax, 0.99 * xmax)
    res = {}
    for fb in (False, None):
        s = capi.Solver(12, 4, 30, b, structured_fallback=fb)
        s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=-xmax, xmax=xmax, rho=30.0, rho_profile="stiffness", terminal=terminal)
        s.set_reference(p.x_ref, p.u_ref); s.update_initialization(X0)
        o = capi.default_opts(rho=30.0, max_iter=8, check_every=8)
        for _ in range(2): s.calculate(o)
        t0 = time.perf_counter()
        for _ in range(5): s.calculate(o)
        el = (time.perf_counter() - t0) / 5
        res[fb] = (s.get_results(want=("status", "polish_iters", "u")), el)
        if os.environ.get("STAMPS") and fb is None:
            L = s.L
            L.almpc_dbg_stamps_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.almpc_dbg_stamps_fetch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
            assert L.almpc_dbg_stamps_enable(s.h, 4 * b) == 0
            s.calculate(o)
            stamps = np.zeros((4 * b, 16), dtype=np.int64); assert L.almpc_dbg_stamps_fetch(s.h, stamps.ctypes.data, 4 * b) == 0
        s.close()
    bad = np.flatnonzero(res[False][0]["status"] == 1)
    print(f"case {case} (amp {amp} box x{box} terminal {terminal}): finish alone {1e6 * res[False][1]:.0f} us, with the redo {1e6 * res[None][1]:.0f} us; "
          f"{len(bad)} left by the finish")
    if os.environ.get("STAMPS"):
        names = ["sweeps: backward", "sweeps: forward", "scan", "Sinv c, sources / columns", "ratio test, step", "border / remove"]
        w0 = stamps[bad, 12].min()
        print("   wall clock (us from the first body start): " + ", ".join(f"{i}: {(stamps[i, 12] - w0) / 100:.0f}..{(stamps[i, 13] - w0) / 100:.0f} ({int(stamps[i, 6])} cyc)" for i in bad))
        for i in bad[:6]:
            print(f"   instance {i}: k_sdual cycles " + ", ".join(f"{nm} {int(stamps[i, c])}" for c, nm in enumerate(names)) + f"; changes {stamps[i, 8]}, bw stages {stamps[i, 9]}, fw stages {stamps[i, 10]}, rows {stamps[i, 11]}")
    for i in bad[:16]:
        u = res[None][0]["u"][i]
        nact = int(((u >= p.u_max[:, None] - 1e-12) | (u <= p.u_min[:, None] + 1e-12)).sum())
        print(f"   instance {i}: finish iterations {res[False][0]['polish_iters'][i]}, redo status {res[None][0]['status'][i]} after {res[None][0]['polish_iters'][i]} changes, inputs on a bound {nact}")
