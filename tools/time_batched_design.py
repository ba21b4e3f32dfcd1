"""Time almpc_design_batched for 4096 quadrotor-size models (the bench's per_instance_models.design_ms) and the nz = 100 / 256 and
nz = 40 / 1024 shapes: python tools/time_batched_design.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
q = mo.quadrotor()
for name, n, m, N, b in (("quadrotor 12-4-30 x 4096", 12, 4, 30, 4096), ("4-2-50 x 256", 4, 2, 50, 256), ("4-2-20 x 1024", 4, 2, 20, 1024)):
    if n == 12:
        A = np.repeat(q.A[None], b, 0); B = q.B[None] * (1.0 + 0.05 * np.sin(np.arange(b)))[:, None, None]; P = q.P; Q, R = q.Q, q.R
        umin, umax = q.u_min, q.u_max
    else:
        rng = np.random.default_rng(1)
        A = np.eye(n)[None] * 0.9 + 0.05 * rng.standard_normal((b, n, n)); B = 0.3 * rng.standard_normal((b, n, m))
        Q, R, P = 100.0 * np.eye(n), 0.1 * np.eye(m), 150.0 * np.eye(n); umin, umax = -np.ones(m), np.ones(m)
    s = capi.Solver(n, m, N, b)
    s.design_batched(A, B, Q, R, None, P, umin, umax)
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); s.design_batched(A, B, Q, R, None, P, umin, umax); t.append(time.perf_counter() - t0)
    H0, F0, d0 = s.get_design_instance(b - 1)
    s.close()
    print(f"{name}: design_batched {1e3 * min(t):.2f} ms (best of 5)")
