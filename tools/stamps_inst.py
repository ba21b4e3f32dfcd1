"""Diagnostic (-DALMPC_STAMPS build): phase cycles of k_admm_inst for the first two instances of every workgroup."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ALMPC_LIB"] = os.path.join(ROOT, "automationlabsmodelpredictivecontrol.jl_amd", "lib", "libalmpc_stamps.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
b = 4096; K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = mo.quadrotor()
As = np.repeat(p.A[None], b, 0); Bs = np.repeat(p.B[None], b, 0)
s = capi.Solver(12, 4, 30, b)
s.design_batched(As, Bs, p.Q, p.R, None, p.P, p.u_min, p.u_max, rho=30.0, rho_profile="stiffness")
s.update_initialization(mo.quadrotor_x0_batch(b, 1.0))
opts = capi.default_opts(rho=30.0, max_iter=K, check_every=K, polish=0)
for _ in range(3): s.calculate(opts)
W = 3 * b
s.L.almpc_dbg_stamps_enable(s.h, W)
s.calculate(opts)
out = np.zeros((W, 16), dtype=np.int64)
s.L.almpc_dbg_stamps_fetch(s.h, out.ctypes.data_as(ctypes.c_void_p), W)
t = out[:256]
names = ["loads (M, F, V, constants) + barrier", "F e0 / V e0 partials + barrier + sums", "init", "-", f"{K} iterations", "hand-off stores"]
for ordn in (0, 1):
    d = np.diff(t[:, ordn * 8: ordn * 8 + 7], axis=1)
    print("instance", ordn, "of each workgroup (cycles, median / max):")
    for i, nm in enumerate(names):
        print(f"   {nm:42s} {int(np.median(d[:, i])):8d} {int(d[:, i].max()):8d}")
print("gap between end of instance 0 and top of instance 1:", int(np.median(t[:, 8] - t[:, 6])))
s.close()
