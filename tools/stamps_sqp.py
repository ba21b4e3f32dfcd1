"""Diagnostic: per-phase cycle stamps of k_polish<false> inside one SQP iteration (needs the -DALMPC_STAMPS build, see tools/README.md)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ALMPC_LIB"] = os.path.join(ROOT, "automationlabsmodelpredictivecontrol.jl_amd", "lib", "libalmpc_stamps.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
bq, nq, mq, Nq = 256, 4, 2, 50
f = mo.synthetic_fnn(act="tanh")
xr = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, Nq + 1)); ur = np.tile(np.array([0.1, -0.2])[:, None], (1, Nq))
X0 = xr[:, 0][None, :] + 0.6 * mo.splitmix_normal(0x5EED0005, 0, bq, nq)
Al, Bl = f.jacobian(xr[:, -1], ur[:, -1]); P = mo.dare(Al, Bl, 100.0 * np.eye(nq), 0.1 * np.eye(mq))
s = capi.Solver(nq, mq, Nq, bq)
s.sqp_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, xr, ur, 100.0 * np.eye(nq), 0.1 * np.eye(mq), None, P, -np.ones(mq), np.ones(mq), act="tanh")
s.sqp_fnn_start(X0)
s.sqp_fnn_iterate(6)
W = 3 * 4096
s.L.almpc_dbg_stamps_enable(s.h, W)
s.sqp_fnn_iterate(1)
out = np.zeros((W, 16), dtype=np.int64)
s.L.almpc_dbg_stamps_fetch(s.h, out.ctypes.data_as(ctypes.c_void_p), W)
o = out[:bq]
ok = o[:, 8] > 0
names = ["prologue(loads)", "initial set+GJ+recompute", "active-set loop", "w/u/eu"]
d = np.diff(o[ok][:, [8, 9, 10, 11, 12]], axis=1)
for i, nm in enumerate(names):
    v = d[:, i]
    print(f"{nm:28s} cycles: median {int(np.median(v)):7d}  p90 {int(np.percentile(v,90)):7d}  max {v.max():7d}")
g = np.diff(o[:, 0:6], axis=1)
for i, nm in enumerate(["guess", "flags + positions", "gather K", "Gauss-Jordan", "write-out"]):
    print(f"k_guess_iterate_ws {nm:20s} cycles: median {int(np.median(g[:, i])):7d}  max {g[:, i].max():7d}")
print("k_guess_iterate_ws inside the sweep: publication + barrier", int(np.median(o[:, 13])), " reads of the pivot rows", int(np.median(o[:, 14])))
print("k_guess_iterate_ws wall clock us: median", np.median(o[:, 7] - o[:, 6]) / 100, " cycles per us:", np.median((o[:, 5] - o[:, 0]) / np.maximum(1, (o[:, 7] - o[:, 6]) / 100)))
print("instances stamped", int(ok.sum()), " total median", int(np.median(o[ok][:, 12] - o[ok][:, 8])), "max", int((o[ok][:, 12] - o[ok][:, 8]).max()))
u = s.get_results(want=('u',))['u']
na = (np.abs(u) >= 1.0).reshape(bq, -1).sum(axis=1)
print('inputs on a bound per instance: median', int(np.median(na)), 'min', na.min(), 'max', na.max())
s.close()
