"""Diagnostic: run the -DALMPC_STAMPS build and print per-phase cycle statistics (lane-0 stamps of every wave)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ALMPC_LIB"] = os.environ.get("ALMPC_STAMPS_LIB") or os.path.join(ROOT, "automationlabsmodelpredictivecontrol.jl_amd", "lib", "libalmpc_stamps.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, bench, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
amp = sys.argv[1] if len(sys.argv) > 1 else "mix"
p = mo.quadrotor(); B = 4096
X0 = bench.make_x0(mo, 0, B, None if amp == "mix" else float(amp))
prof = os.environ.get('ALMPC_RHO_PROFILE', 'scalar'); rho = float(os.environ.get('ALMPC_RHO', '0.1'))
s = capi.Solver(12, 4, 30, B); s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, rho=rho, rho_profile=prof); s.set_reference(p.x_ref, p.u_ref); s.update_initialization(X0)
opts = capi.default_opts(rho=rho, max_iter=int(sys.argv[2]) if len(sys.argv) > 2 else 25, check_every=int(sys.argv[2]) if len(sys.argv) > 2 else 25)
for _ in range(3): s.calculate(opts)
L = s.L; W = 3 * B
L.almpc_dbg_stamps_enable(s.h, W)
s.calculate(opts)
out = np.zeros((W, 16), dtype=np.int64)
L.almpc_dbg_stamps_fetch(s.h, out.ctypes.data_as(ctypes.c_void_p), W)
pit = s.get_results(want=("polish_iters",))["polish_iters"]
full = out
out = full[:B]
sl = [8, 9, 10, 11, 12, 13, 14]
names = ["prologue(loads)", "initial set+GJ+recompute", "active-set loop", "w/u/eu + Z fill", "rollout steps", "x/ex out"]
ok = out[:, 8] > 0
d = np.diff(out[ok][:, sl], axis=1)
for i, nm in enumerate(names):
    v = d[:, i]
    print(f"{nm:28s} cycles: median {int(np.median(v)):7d}  p90 {int(np.percentile(v,90)):7d}  max {v.max():7d}   mean {v.mean():9.0f}")
ok2 = (out[:, 0] > 0) & (out[:, 4] > 0) & (np.arange(B) < B // 16 * 8)
da = np.diff(out[ok2][:, [0, 1, 2, 3, 4]], axis=1)
for i, nm in enumerate(["admm prologue", "admm loop", "admm v0 product", "admm result flush"]):
    v = da[:, i]
    print(f"{nm:28s} cycles: median {int(np.median(v)):7d}  p90 {int(np.percentile(v,90)):7d}  max {v.max():7d}")
ex = out[ok2]
print("admm prologue split: start->e0 written", int(np.median(ex[:,5]-ex[:,0])), " barrier", int(np.median(ex[:,6]-ex[:,5])), " f' + constants", int(np.median(ex[:,7]-ex[:,6])), " init + nf reduce", int(np.median(ex[:,1]-ex[:,7])))
fine = full[10240:10240 + 2048]
okf = fine[:, 0] > 0
if okf.any():
    base = out[:2048][okf][:, 0]
    print("admm prologue fine (cycles from the wave's start): x0 loads issued", int(np.median(fine[okf, 0] - base)), " + F' / constants issued",
          int(np.median(fine[okf, 1] - base)), " + fragments issued", int(np.median(fine[okf, 2] - base)), " e0 in LDS", int(np.median(out[:2048][okf][:, 5] - base)))
print("admm loop cycles / iteration:", np.median(da[:, 1]) / opts.max_iter)
tot = out[ok][:, 14] - out[ok][:, 8]
print("total per wave: median", int(np.median(tot)), "p90", int(np.percentile(tot, 90)), "max", tot.max(), " (2.4 GHz: max = %.1f us)" % (tot.max() / 2400.0))
loop = d[:, 2]; its = pit[ok]
print("loop cycles per polish iteration (instances with >= 5 its): median", int(np.median(loop[its >= 5] / its[its >= 5])))
acc = full[4096:4096 + B]
n = acc[:, 8]
sel = n >= 8
print("per-add phase cycles (instances with >= 8 adds in the loop), mean per add:")
for i, nm in enumerate(["A loads+rowbuf+gather c", "B put c + Sinv matvec", "C put u + G rows fma", "D sum/div/t update", "E rank1+border+sync", "F ratio test"]):
    print(f"   {nm:26s} {np.mean(acc[sel, i] / n[sel]):8.0f}")
s.close()
# the slowest waves in detail
okidx = np.nonzero(ok)[0]
order = np.argsort(-tot)[:6]
print("slowest instances: inst, total, per-phase cycles, polish its, adds counted")
for o in order:
    i = okidx[o]
    print("  ", i, int(tot[o]), d[o].tolist(), int(pit[i]), int(acc[i, 8]), "acc", (acc[i, :6]).tolist())

tl = full[8192:8192 + 2048]
okw = tl[:, 0] > 0
if okw.any():
    gl = tl[okw, 1] - tl[okw, 0]; tt = tl[okw, 2] - tl[okw, 0]; cnt = tl[okw, 3]
    print("persistent waves:", int(okw.sum()), " G load cycles median", int(np.median(gl)), "max", int(gl.max()),
          " total cycles median", int(np.median(tt)), "max", int(tt.max()), " instances per wave min/median/max", int(cnt.min()), int(np.median(cnt)), int(cnt.max()))
    print("kernel span over all waves (if clocks comparable):", int(tl[okw, 2].max() - tl[okw, 0].min()))
if okw.any():
    # per workgroup (8 waves): when does its last wave finish the polish phase, relative to the phase start of that workgroup
    t0 = tl[okw, 1].reshape(-1, 8).min(axis=1); t1 = tl[okw, 2].reshape(-1, 8).max(axis=1)
    dur = t1 - t0
    print("polish phase per workgroup (cycles): min", int(dur.min()), "median", int(np.median(dur)), "p90", int(np.percentile(dur, 90)), "max", int(dur.max()))
