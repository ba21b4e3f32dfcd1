"""Per-iteration vs fixed cost of k_admm_inst: python tools/time_batched_k.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
b = int(os.environ.get("ALMPC_BATCH", "4096"))
if os.environ.get("ALMPC_SHAPE") == "c4":
    p = mo.make_problem(0.9 * np.eye(4) + 0.05 * np.ones((4, 4)), np.ones((4, 2)) * 0.3 + np.eye(4)[:, :2], 20, [-1, -1], [1, 1])
    X0 = mo.splitmix_normal(1, 0, b, 4)
else:
    p = mo.quadrotor()
    X0 = mo.quadrotor_x0_batch(b, 1.0)
As = np.repeat(p.A[None], b, 0); Bs = np.repeat(p.B[None], b, 0)
s = capi.Solver(p.n, p.m, p.N, b, timing=True)
s.design_batched(As, Bs, p.Q, p.R, None, p.P, p.u_min, p.u_max, rho=30.0, rho_profile="stiffness")
s.update_initialization(X0)
for K in (1, 2, 4, 8, 16, 32):
    opts = capi.default_opts(rho=30.0, max_iter=K, check_every=K, polish=0)
    for _ in range(3): s.calculate(opts)
    s.timing_reset(20)
    for _ in range(20): s.calculate(opts, sync=False)
    s.synchronize()
    ts = s.timing_summary()
    print(f"K={K:3d}: admm {1e3*ts['admm_ms']/ts['steps']:.1f} us")
s.close()
