"""Diagnostic: where a k_riccati sweep spends its cycles (-DALMPC_STAMPS build): python tools/stamps_riccati.py"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ALMPC_LIB"] = os.path.join(ROOT, "automationlabsmodelpredictivecontrol.jl_amd", "lib", "libalmpc_stamps.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
q = mo.quadrotor(N=50); b = 1024
X0 = mo.quadrotor_x0_batch(b, 1.0)
s = capi.Solver(12, 4, 50, b, structured=True)
s.design_shared(q.A, q.B, q.Q, q.R, None, None, q.u_min, q.u_max)
s.update_initialization(X0)
s.calculate()
L = s.L
L.almpc_dbg_stamps_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.almpc_dbg_stamps_fetch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
assert L.almpc_dbg_stamps_enable(s.h, b) == 0
s.calculate()
out = np.zeros((b, 16), dtype=np.int64); assert L.almpc_dbg_stamps_fetch(s.h, out.ctypes.data, b) == 0
names = ["P+A, P+B (two products)", "B'P+A, B'P+B, g", "Lam, h, masks", "Lam^-1 (m pivots)", "K, kff, store", "P update + store", "forward rollout"]
its = out[:, 8]
tot = out[:, :7].sum(axis=1)
print("instances", b, "iterations median", int(np.median(its)), "max", its.max(), "cycles in sweeps: median %d max %d" % (np.median(tot), tot.max()))
w = out[:, :7].sum(axis=0)
for i, nm in enumerate(names): print(f"   {nm:28s} {100.0 * w[i] / w.sum():5.1f} %")
stages = 50.0 * its   # upper bound on backward stages (partial sweeps do fewer)
print("cycles per (iteration x 50 stages), median: %.0f" % np.median(tot / np.maximum(stages, 1)))
s.close()
