"""Diagnostic: per-phase cycles of the state-row finish (k_polish_gen / k_polish_gen64) from the -DALMPC_STAMPS build.
   python tools/stamps_state_rows.py <case 0..5>   (cases of tools/time_state_rows.py: (amp, box) x terminal)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ALMPC_LIB"] = os.path.join(ROOT, "automationlabsmodelpredictivecontrol.jl_amd", "lib", "libalmpc_stamps.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ctypes
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
p = mo.quadrotor(); b = 4096
case = int(sys.argv[1]) if len(sys.argv) > 1 else 0
amp, box = ((1.0, 3.0), (1.0, 1.0), (3.0, 3.0))[case // 2]
terminal = ("none", "equality")[case % 2]
X0 = mo.quadrotor_x0_batch(b, amp)
xmax = box * np.array([1, 1, 1, .5, .5, .5, .1, .1, .1, .1, .1, .1]) * 3.0
X0 = np.clip(X0, -0.99 * xmax, 0.99 * xmax)
s = capi.Solver(12, 4, 30, b)
s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=-xmax, xmax=xmax, rho=30.0, rho_profile="stiffness", terminal=terminal)
s.set_reference(p.x_ref, p.u_ref); s.update_initialization(X0)
o = capi.default_opts(rho=30.0, max_iter=8, check_every=8)
for _ in range(2): s.calculate(o)
L = s.L
L.almpc_dbg_stamps_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.almpc_dbg_stamps_fetch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
W = 4 * b
assert L.almpc_dbg_stamps_enable(s.h, W) == 0
s.calculate(o)
out = np.zeros((W, 16), dtype=np.int64)
assert L.almpc_dbg_stamps_fetch(s.h, out.ctypes.data, W) == 0
r = s.get_results(want=("status", "polish_iters"))
print(f"amp {amp} box x{box} terminal {terminal}: status {np.bincount(r['status'], minlength=4).tolist()}")
names = ["load + s0 rollout + rows", "guess build", "recompute", "dual feasibility", "main loop", "outputs + rollout"]
for tier, sl in (("first launch (32 rows)", slice(0, b)), ("second launch (64 rows)", slice(b, 2 * b))):
    st = out[sl]
    on = st[:, 0] != 0
    if not on.any():
        print(tier, ": no instances"); continue
    st = st[on]
    print(f"{tier}: {on.sum()} instances; kernel span {st[:, 6].max() - st[:, 0].min()} cycles")
    for j, nm in enumerate(names):
        d = st[:, j + 1] - st[:, j]
        print(f"   {nm:28s} median {int(np.median(d)):7d}  p90 {int(np.percentile(d, 90)):7d}  max {int(d.max()):7d}")
    for nm, a, b_ in (("  prologue: loads + Z fill", 0, 11), ("  prologue: rollout of v0", 11, 12), ("  prologue: row data", 12, 1),
                      ("  epilogue: u / e_u stores", 5, 13), ("  epilogue: rollout", 13, 14), ("  epilogue: x / e_x stores", 14, 6)):
        d = st[:, b_] - st[:, a]
        print(f"   {nm:28s} median {int(np.median(d)):7d}  p90 {int(np.percentile(d, 90)):7d}")
    tot = st[:, 6] - st[:, 0]
    print(f"   total per instance          median {int(np.median(tot)):7d}  p90 {int(np.percentile(tot, 90)):7d}  max {int(tot.max()):7d}")
    its = st[:, 9]; kg = st[:, 8]; kf = st[:, 10]
    ml = st[:, 5] - st[:, 4]
    print(f"   guess rows median {int(np.median(kg))} max {kg.max()}; final rows median {int(np.median(kf))} max {kf.max()}; iterations median {int(np.median(its))} max {its.max()}")
    big = its >= 5
    if big.any():
        print(f"   main-loop cycles per iteration (instances with >= 5): median {int(np.median(ml[big] / its[big]))}")
    acc = out[2 * b:][sl][on]
    tot_ml = max(1, int(ml.sum()))
    for j, nm in enumerate(("scan + selection", "dir_u (row j, c, u = Sinv c, dp)", "dir_d (Ghat[W,:]' u)", "step lengths", "partial step + removal", "full step + border", "confirmation")):
        print(f"   main loop: {nm:34s} {100.0 * acc[:, j].sum() / tot_ml:5.1f} % of the main-loop cycles")
    gb = (st[:, 2] - st[:, 1])
    nzk = kg > 0
    if nzk.any():
        print(f"   guess-build cycles per row: median {int(np.median(gb[nzk] / kg[nzk]))}")
s.close()
