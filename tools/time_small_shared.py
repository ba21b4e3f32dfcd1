"""Step time of small shared-model problems (the reference's own test size: QTP fixture n 4, m 2, N 5; and N 20) by batch size, with the
one-wave-per-instance step (k_step_inst_wave on shared operands) and with the two-launch path (ALMPC_NO_SHARED_WAVE=1):
python tools/time_small_shared.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, almpc_loader, importlib
pkg = almpc_loader.load_package(); capi = pkg._capi
wl = importlib.import_module(pkg.__name__ + ".workloads")
for N in (5, 20):
    p = wl.qtp_fixture(N)
    for b in (1, 64, 512, 2048, 4096, 16384, 65536):
        X = 0.65 + 0.25 * wl.splitmix_normal(0x5EED0051, 0, b, 4)
        row = []
        for env in ("0", "1", "force"):
            os.environ.pop("ALMPC_SHARED_WAVE_MAX_BATCH", None)
            os.environ["ALMPC_NO_SHARED_WAVE"] = "1" if env == "1" else "0"
            if env == "force": os.environ["ALMPC_SHARED_WAVE_MAX_BATCH"] = "1000000000"
            s = capi.Solver(p.n, p.m, p.N, b)
            s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max)
            s.set_reference(p.x_ref, p.u_ref); s.update_initialization(X)
            for _ in range(5): s.calculate()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(50): s.calculate(sync=False)
                s.synchronize()
                best = min(best, (time.perf_counter() - t0) / 50)
            s.close()
            row.append(best)
        print(f"N {N} batch {b:6d}: default {1e6*row[0]:8.1f} us   two-launch {1e6*row[1]:8.1f} us   one-wave forced {1e6*row[2]:8.1f} us", flush=True)
os.environ.pop("ALMPC_NO_SHARED_WAVE", None); os.environ.pop("ALMPC_SHARED_WAVE_MAX_BATCH", None)
