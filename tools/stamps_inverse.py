"""Diagnostic: cycles between consecutive pivots of k_design_inverse_t (thread 0 of the first and of the middle workgroup), for a
batch that fills the GPU and for one that does not (-DALMPC_STAMPS build): python tools/stamps_inverse.py"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ALMPC_LIB"] = os.path.join(ROOT, "automationlabsmodelpredictivecontrol.jl_amd", "lib", "libalmpc_stamps.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
q = mo.quadrotor()
for b in (64, 4096):
    A = np.repeat(q.A[None], b, 0); B = q.B[None] * (1.0 + 0.05 * np.sin(np.arange(b)))[:, None, None]
    s = capi.Solver(12, 4, 30, b)
    s.design_batched(A, B, q.Q, q.R, None, q.P, q.u_min, q.u_max)
    L = s.L
    L.almpc_dbg_stamps_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.almpc_dbg_stamps_fetch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    assert L.almpc_dbg_stamps_enable(s.h, 64) == 0
    s.design_batched(A, B, q.Q, q.R, None, q.P, q.u_min, q.u_max)
    out = np.zeros((64, 16), dtype=np.int64); assert L.almpc_dbg_stamps_fetch(s.h, out.ctypes.data, 64) == 0
    flat = out.reshape(-1)
    for w, nm in ((0, "first workgroup"), (1, "middle workgroup")):
        t = flat[w * 256: w * 256 + 120]
        d = np.diff(t)
        print(f"batch {b:5d} {nm:17s}: cycles per pivot median {int(np.median(d))} min {d.min()} max {d.max()}  (120 pivots: {t[-1] - t[0]} cycles)")
    seg = flat[2 * 256: 2 * 256 + 5]
    print("          thread 0 of the first workgroup, cycles per pivot: publish %d | barrier %d | pivot read + reciprocal %d | row reads %d | column reads + FMAs %d"
          % tuple(int(v / 120) for v in seg))
    s.close()
