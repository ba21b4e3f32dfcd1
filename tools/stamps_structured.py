"""Diagnostic: where k_sdual spends its cycles (-DALMPC_STAMPS build): python tools/stamps_structured.py [N] [case]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ALMPC_LIB"] = os.path.join(ROOT, "automationlabsmodelpredictivecontrol.jl_amd", "lib", "libalmpc_stamps.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
case = sys.argv[2] if len(sys.argv) > 2 else "box-only"
XMAX = np.array([3, 3, 3, 1.5, 1.5, 1.5, 0.3, 0.3, 0.3, 1.0, 1.0, 1.0])
kw = {"box-only": {}, "box": dict(x_min=-XMAX, x_max=XMAX), "eq": dict(terminal="equality"), "S": dict(s=5.0)}[case]
q = mo.quadrotor(N); b = 4096
p = mo.make_problem(q.A, q.B, N, q.u_min, q.u_max, **kw)
X0 = np.concatenate([mo.quadrotor_x0_batch(b // 4 if a != 1.0 else b // 2, a, first_instance=k * b) for k, a in enumerate((0.3, 1.0, 3.0))])[:b]
if p.x_min is not None: X0 = np.clip(X0, -0.99 * XMAX, 0.99 * XMAX)
s = capi.Solver(12, 4, N, b, structured=True)
s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=p.x_min, xmax=p.x_max, terminal=p.terminal)
s.update_initialization(X0)
s.calculate()
L = s.L
L.almpc_dbg_stamps_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.almpc_dbg_stamps_fetch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
assert L.almpc_dbg_stamps_enable(s.h, b) == 0
s.calculate()
out = np.zeros((b, 16), dtype=np.int64); assert L.almpc_dbg_stamps_fetch(s.h, out.ctypes.data, b) == 0
names = ["sweeps: backward", "sweeps: forward", "scan (most violated row)", "Sinv c, sources", "ratio test, step", "border / remove"]
its, nbw, nfw = out[:, 8], out[:, 9], out[:, 10]
tot = out[:, :6].sum(axis=1)
print(f"N {N} {case}: instances {b}, iterations median {int(np.median(its))} max {its.max()}, stamped cycles median {int(np.median(tot))} max {tot.max()}")
w = out[:, :6].sum(axis=0)
for i, nm in enumerate(names): print(f"   {nm:28s} {100.0 * w[i] / w.sum():5.1f} %")
print("cycles per backward stage: %.0f   per forward stage: %.0f" % (out[:, 0].sum() / max(nbw.sum(), 1), out[:, 1].sum() / max(nfw.sum(), 1)))
j = int(np.argmax(tot)); print("slowest instance", j, "its", its[j], "rows", out[j, 11], "cycles", out[j, :6].tolist())
s.close()
