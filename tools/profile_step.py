"""Run K MPC steps of the headline workload (for rocprofv3): python tools/profile_step.py [amplitude|mix] [steps] [max_iter]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import almpc_loader, bench
import mpc_oracle as mo
capi = almpc_loader.load_package()._capi
amp = sys.argv[1] if len(sys.argv) > 1 else "mix"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
p = mo.quadrotor()
B = int(os.environ.get("ALMPC_BATCH", "4096"))
X0 = bench.make_x0(mo, 0, B, None if amp == "mix" else float(amp))
s = capi.Solver(12, 4, 30, B, timing=True)
prof = os.environ.get('ALMPC_RHO_PROFILE', 'scalar'); rho = float(os.environ.get('ALMPC_RHO', '0.1'))
s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, rho=rho, rho_profile=prof)
s.set_reference(p.x_ref, p.u_ref)
s.update_initialization(X0)
opts = capi.default_opts(rho=rho) if len(sys.argv) <= 3 else capi.default_opts(rho=rho, max_iter=int(sys.argv[3]), check_every=int(sys.argv[4]) if len(sys.argv) > 4 else 25)
for _ in range(5):
    s.calculate(opts)
s.timing_reset(steps)
for _ in range(steps):
    s.calculate(opts, sync=False)
s.synchronize()
t = s.timing_summary()
r = s.get_results(want=("status", "iters", "polish_iters"))
print({k: (v / t["steps"] if k != "steps" else v) for k, v in t.items()}, "polish iters mean", r["polish_iters"].mean(), "max", r["polish_iters"].max(), "status", np.bincount(r["status"]))
s.close()
