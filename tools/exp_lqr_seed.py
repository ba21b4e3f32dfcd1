"""CPU experiment (numpy restatement, round 5): does a SATURATED CLOSED-LOOP LQR ROLLOUT give the exact active-set finish a better
working-set seed than ADMM's sign(y) guess?  (Round-4 review, item 1.)

With P = DARE the unconstrained optimum of the reference's QP (src/sub/design_mpc.jl:327, :405-468) is the LQR feedback
u_k = -K e_k; rolling the plant forward under u_k = clip(-K e_k) (the clipping propagated through the state) gives a FEASIBLE
input sequence whose saturated rows are the seed.  For the benchmark batch (4096 quadrotor instances, amplitudes 0.3/1/3 interleaved)
this prints, per seed variant: mean / max working-set changes of the finish (adds + removes + purged rows), wrong rows in the seed,
missing rows, and the two named hard instances 1613 / 1871.

    python tools/exp_lqr_seed.py [batch] [procs]
"""
import os
import sys
from multiprocessing import Pool

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import mpc_oracle as mo
import bench

p = mo.quadrotor()
des = mo.design_shared(p, rho=45.0, sigma=1e-6, rho_profile="stiffness")
G, lo, hi, d = des["G"], des["lo"], des["hi"], des["d"]
nz = lo.size
Klqr = np.linalg.solve(p.R + p.B.T @ p.P @ p.B, p.B.T @ p.P @ p.A)
ULO = p.u_min[:, None] - p.u_ref   # (m, N) bounds on v = e_u
UHI = p.u_max[:, None] - p.u_ref


def lqr_sat(e0):
    """u_k = clip(-K e_k), e_{k+1} = A e_k + B u_k.  Returns v (N*m, unscaled) and side (N*m: -1 / 0 / +1)."""
    e = e0.copy()
    v = np.empty((p.N, p.m))
    side = np.zeros((p.N, p.m), dtype=int)
    for k in range(p.N):
        uk = -Klqr @ e
        side[k] = np.where(uk < ULO[:, k], -1, np.where(uk > UHI[:, k], 1, 0))
        uk = np.clip(uk, ULO[:, k], UHI[:, k])
        v[k] = uk
        e = p.A @ e + p.B @ uk
    return v.reshape(-1), side.reshape(-1)


def finish(v0, w_start, seed_side, purge=True, max_iter=600):
    """polish_active_set of oracle/mpc_oracle.py with an explicit seed (side per row) and feasible start w_start."""
    W, side = [], {}
    Sinv = np.zeros((0, 0))

    def add(j, s):
        nonlocal Sinv
        c = G[W, j]
        u = Sinv @ c
        sc = G[j, j] - c @ u
        k = len(W)
        Sn = np.empty((k + 1, k + 1))
        Sn[:k, :k] = Sinv + np.outer(u, u) / sc
        Sn[:k, k] = -u / sc
        Sn[k, :k] = -u / sc
        Sn[k, k] = 1.0 / sc
        Sinv = Sn
        W.append(j)
        side[j] = s

    def remove(pos):
        nonlocal Sinv
        keep = [i for i in range(len(W)) if i != pos]
        Sinv = Sinv[np.ix_(keep, keep)] - np.outer(Sinv[keep, pos], Sinv[pos, keep]) / Sinv[pos, pos]
        del side[W.pop(pos)]

    w = np.clip(w_start, lo, hi)
    for j in range(nz):
        if seed_side[j] < 0:
            w[j] = lo[j]
            add(j, -1)
        elif seed_side[j] > 0:
            w[j] = hi[j]
            add(j, +1)
    k0 = len(W)
    n_add = n_rem = n_purged = 0
    if purge and 0 < len(W):
        b = np.array([hi[j] if side[j] > 0 else lo[j] for j in W])
        lam = Sinv @ (v0[W] - b)
        viol = np.array([-lam[i] if side[W[i]] > 0 else lam[i] for i in range(len(W))])
        bad = viol > 1e-12 * max(1.0, float(np.max(np.abs(lam))))
        if bad.sum() >= 2:
            kept = [(W[i], side[W[i]]) for i in range(len(W)) if not bad[i]]
            n_purged = int(bad.sum())
            W, side, Sinv = [], {}, np.zeros((0, 0))
            for j, sd in kept:
                add(j, sd)
    for it in range(1, max_iter + 1):
        if W:
            b = np.array([hi[j] if side[j] > 0 else lo[j] for j in W])
            r = v0[W] - b
            lam = Sinv @ r
            lam = lam + Sinv @ (r - G[np.ix_(W, W)] @ lam)
            t = v0 - G[:, W] @ lam
            t[W] = b
        else:
            lam = np.zeros(0)
            t = v0.copy()
        step = t - w
        free = np.ones(nz, dtype=bool)
        free[W] = False
        with np.errstate(divide="ignore", invalid="ignore"):
            r_hi = np.where(free & (t > hi), (hi - w) / step, np.inf)
            r_lo = np.where(free & (t < lo), (lo - w) / step, np.inf)
        rr = np.minimum(r_hi, r_lo)
        j = int(np.argmin(rr))
        if rr[j] < 1.0:
            w = w + max(rr[j], 0.0) * step
            if r_hi[j] <= r_lo[j]:
                w[j] = hi[j]
                add(j, +1)
            else:
                w[j] = lo[j]
                add(j, -1)
            n_add += 1
            continue
        w = t
        if not W:
            break
        viol = np.array([-lam[i] if side[W[i]] > 0 else lam[i] for i in range(len(W))])
        i = int(np.argmax(viol))
        if viol[i] <= 1e-12 * max(1.0, float(np.max(np.abs(lam)))):
            break
        remove(i)
        n_rem += 1
    fin = np.zeros(nz, dtype=int)
    for j in W:
        fin[j] = side[j]
    return dict(w=np.clip(w, lo, hi), k0=k0, n_add=n_add, n_rem=n_rem, n_purged=n_purged, fin=fin)


VARIANTS = ["admm6", "admm3", "admm1", "lqr", "lqr|admm6", "lqr&admm6", "lqr|admm3", "lqr|admm1", "lqr&admm3"]


def admm_guess(fs, K):
    r = mo.admm_box(des["Hs"], fs, lo, hi, rho=des["rho_vec"], sigma=des["sigma"], alpha=1.6, max_iter=K, check_every=K,
                    Minv=des["Minv"], unscale=d)
    w = np.clip(r["z"], lo, hi)
    s = np.where((r["y"] < 0) & (w <= lo), -1, np.where((r["y"] > 0) & (w >= hi), 1, 0))
    return w, s


def one(args):
    i, x0 = args
    e0 = x0 - p.x_ref[:, 0]
    fs = des["Fs"] @ e0 + des["fS"]
    v0 = -G @ fs
    vl, sl = lqr_sat(e0)
    wl = vl / d
    g = {K: admm_guess(fs, K) for K in (6, 3, 1)}
    out = {}
    ref = None
    for name in VARIANTS:
        if name.startswith("admm"):
            w0, s0 = g[int(name[4:])]
        elif name == "lqr":
            w0, s0 = wl, sl
        else:
            K = int(name[-1])
            wa, sa = g[K]
            if "|" in name:   # union: LQR's side wins where both name a row; start = the LQR rollout (feasible) with ADMM's rows put on their bound
                s0 = np.where(sl != 0, sl, sa)
            else:
                s0 = np.where(sl == sa, sl, 0)
            w0 = wl
        r = finish(v0, w0, s0)
        if ref is None:
            ref = r
        else:
            assert np.max(np.abs(r["w"] - ref["w"])) < 1e-6, (i, name, np.max(np.abs(r["w"] - ref["w"])))
        wrong = int(np.sum((s0 != 0) & (s0 != r["fin"])))
        missing = int(np.sum((r["fin"] != 0) & (s0 != r["fin"])))
        out[name] = (r["k0"], r["n_add"], r["n_rem"], r["n_purged"], wrong, missing, int(np.sum(r["fin"] != 0)))
    return i, out


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    X0 = bench.make_x0(mo, 0, B, None)
    with Pool(procs) as pool:
        res = pool.map(one, [(i, X0[i]) for i in range(B)], chunksize=16)
    res.sort()
    print(f"batch {B}: changes = adds + removes (+ purged rows counted once as a rebuild); per seed variant")
    print(f"{'variant':12s} {'mean k0':>8s} {'mean chg':>9s} {'max chg':>8s} {'p99 chg':>8s} {'mean wrong':>11s} {'max wrong':>10s} {'mean miss':>10s} {'max miss':>9s}  i1613(k0,add,rem,purged,wrong,miss,final)  i1871")
    for name in VARIANTS:
        a = np.array([r[1][name] for r in res])
        chg = a[:, 1] + a[:, 2]
        named = "  ".join(str(tuple(int(x) for x in res[i][1][name])) if i < B else "-" for i in (1613, 1871))
        print(f"{name:12s} {a[:, 0].mean():8.2f} {chg.mean():9.2f} {chg.max():8d} {np.percentile(chg, 99):8.1f} {a[:, 4].mean():11.2f} {a[:, 4].max():10d} "
              f"{a[:, 5].mean():10.2f} {a[:, 5].max():9d}  {named}")
    # per class
    for c, amp in enumerate(bench.AMPLITUDES):
        idx = np.arange(B) % 3 == c
        print(f"amplitude {amp}: " + "  ".join(f"{name}: mean {np.array([r[1][name] for r in res])[idx][:, 1:3].sum(1).mean():.2f} max {np.array([r[1][name] for r in res])[idx][:, 1:3].sum(1).max()}" for name in VARIANTS))
