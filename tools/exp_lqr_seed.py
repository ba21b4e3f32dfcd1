"""CPU experiment (numpy restatement, round 5): does a SATURATED CLOSED-LOOP LQR ROLLOUT give the exact active-set finish of the
headline step a better working-set seed than ADMM's sign(y) guess?  (Round-4 review, item 1; result in DESIGN.md section 4: no.)

With P = DARE the unconstrained optimum of the reference's QP (src/sub/design_mpc.jl:327, :405-468) is the LQR feedback; rolling
the plant forward under u_k = clip(-K e_k) (`mpc_oracle.saturated_lqr_rollout`) gives a feasible input sequence whose saturated rows
are the seed of `mpc_oracle.polish_active_set(seed=...)`.  For the benchmark batch (4096 quadrotor instances, amplitudes 0.3 / 1 / 3
interleaved, rho profile 45) this prints

  table 1  per seed variant (ADMM K = 6 / 3 / 1, LQR, unions, intersections): mean / max working-set changes of the finish, rows of
           the seed that are not in the final set ("wrong"), rows of the final set the seed lacks ("missing"), and the two hard
           instances 1613 / 1871 by name;
  table 2  hybrid policies that exploit  sign(y) guess <= final set <= LQR seed : LQR rows added to ADMM's in stage order up to a
           cap, ADMM's rows protected from the purge, one to three purge rounds, LQR distrusted when a channel shows more than two
           arcs (the limit cycle of a saturated LQR loop), with a cycle model of the finish (fixed 24 k, 0.7 k per row of an
           initial inverse, 3.7 k per add, 6 k per remove: tools/stamps.py) and the 8-wave tile schedule.

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
STAGE = np.repeat(np.arange(p.N), p.m)
VARIANTS = ["admm6", "admm3", "admm1", "lqr", "lqr|admm6", "lqr&admm6", "lqr|admm3", "lqr&admm3"]


def admm_guess(fs, K):
    r = mo.admm_box(des["Hs"], fs, lo, hi, rho=des["rho_vec"], sigma=des["sigma"], alpha=1.6, max_iter=K, check_every=K,
                    Minv=des["Minv"], unscale=d)
    w = np.clip(r["z"], lo, hi)
    return w, np.where((r["y"] < 0) & (w <= lo), -1, np.where((r["y"] > 0) & (w >= hi), 1, 0))


def finish_protected(v0, w_start, seed, protect, rounds):
    """The finish with `rounds` purge rounds that never drop protected rows (ADMM's, which are right 4095 times of 4096); a purged
    row goes back to its value in w_start.  Same loop as mo.polish_active_set otherwise (dense solves: counts only)."""
    W = [int(j) for j in np.nonzero(seed)[0]]
    side = {j: int(seed[j]) for j in W}
    w = np.clip(w_start, lo, hi)
    for j in W:
        w[j] = lo[j] if side[j] < 0 else hi[j]
    k0, n_add, n_rem, n_purged = len(W), 0, 0, 0

    def lam_t():
        b = np.array([hi[j] if side[j] > 0 else lo[j] for j in W])
        lam = np.linalg.solve(G[np.ix_(W, W)], v0[W] - b)
        t = v0 - G[:, W] @ lam
        t[W] = b
        return lam, t

    for _ in range(rounds):
        if not W:
            break
        lam, _t = lam_t()
        viol = np.array([-lam[i] if side[W[i]] > 0 else lam[i] for i in range(len(W))])
        bad = (viol > 1e-12 * max(1.0, float(np.max(np.abs(lam))))) & np.array([not protect[j] for j in W])
        if not bad.any():
            break
        n_purged += int(bad.sum())
        for i in sorted(np.nonzero(bad)[0], reverse=True):
            j = W.pop(i)
            del side[j]
            w[j] = min(max(w_start[j], lo[j]), hi[j])
    for _ in range(600):
        lam, t = lam_t() if W else (np.zeros(0), v0.copy())
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
            s = 1 if r_hi[j] <= r_lo[j] else -1
            w[j] = hi[j] if s > 0 else lo[j]
            W.append(j)
            side[j] = s
            n_add += 1
            continue
        w = t
        if not W:
            break
        viol = np.array([-lam[i] if side[W[i]] > 0 else lam[i] for i in range(len(W))])
        i = int(np.argmax(viol))
        if viol[i] <= 1e-12 * max(1.0, float(np.max(np.abs(lam)))):
            break
        del side[W.pop(i)]
        n_rem += 1
    return dict(w=np.clip(w, lo, hi), k0=k0, n_add=n_add, n_rem=n_rem, n_purged=n_purged)


def cycles(r):   # k cycles of one instance's finish (tools/stamps.py: fixed part, initial inverse, purge rebuild, add, remove)
    c = 24.0 + 0.7 * r["k0"] + (0.7 * (r["k0"] - r["n_purged"]) if r["n_purged"] else 0.0)
    return c + 3.7 * r["n_add"] + 6.0 * r["n_rem"]


def arcs(side):   # largest number of sign runs of the saturated rows of one input channel
    worst = 0
    for ch in range(p.m):
        seq = [x for x in side.reshape(p.N, p.m)[:, ch] if x != 0]
        worst = max(worst, sum(1 for i, x in enumerate(seq) if i == 0 or x != seq[i - 1]))
    return worst


def one(args):
    i, x0 = args
    e0 = x0 - p.x_ref[:, 0]
    fs = des["Fs"] @ e0 + des["fS"]
    v0 = -G @ fs
    vl, sl = mo.saturated_lqr_rollout(p, x0)
    wl = vl / d
    g = {K: admm_guess(fs, K) for K in (6, 4, 3, 1)}
    out, ref = {}, None
    for name in VARIANTS:
        if name.startswith("admm"):
            w0, s0 = g[int(name[4:])]
            r = mo.polish_active_set(G, v0, lo, hi, w0, None, seed=s0)
        else:
            if name == "lqr":
                s0 = sl
            else:
                sa = g[int(name[-1])][1]
                s0 = np.where(sl != 0, sl, sa) if "|" in name else np.where(sl == sa, sl, 0)
            r = mo.polish_active_set(G, v0, lo, hi, wl, None, seed=s0)
        if ref is None:
            ref = r
        assert np.max(np.abs(r["w"] - ref["w"])) < 1e-6, (i, name)
        fin = r["side"]
        out[name] = (int((s0 != 0).sum()), r["n_add"], r["n_remove"], r["n_purged"], int(np.sum((s0 != 0) & (s0 != fin))),
                     int(np.sum((fin != 0) & (s0 != fin))), int((fin != 0).sum()))
    pol = {}
    for K in (6, 4):
        wa, sa = g[K]
        base = finish_protected(v0, wa, sa, np.zeros(nz, dtype=bool), 1)
        pol["admm%d" % K] = (base["n_add"] + base["n_rem"], cycles(base))
        for tag, trust, rounds in (("prefix", True, 3), ("arcs<=2", arcs(sl) <= 2 and int(((sl != 0) | (sa != 0)).sum()) <= 32, 3)):
            s = sa.copy()
            if tag == "prefix":   # LQR rows in stage order while the set fits the 32-row register tier
                for k in range(p.N):
                    cand = np.where((STAGE == k) & (sl != 0) & (s == 0))[0]
                    if (s != 0).sum() + len(cand) > 32:
                        break
                    s[cand] = sl[cand]
            elif trust:
                s = np.where(sl != 0, sl, sa)
            r = finish_protected(v0, wa, s, sa != 0, rounds) if trust else base
            assert np.max(np.abs(r["w"] - base["w"])) < 1e-6
            pol["%s K%d" % (tag, K)] = (r["n_add"] + r["n_rem"], cycles(r))
    return i, out, pol


def makespan(c):   # 16 instances of a tile on its 8 waves, hardest first
    t = np.zeros(8)
    for x in sorted(c, reverse=True):
        t[np.argmin(t)] += x
    return t.max()


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    X0 = bench.make_x0(mo, 0, B, None)
    with Pool(procs) as pool:
        res = pool.map(one, [(i, X0[i]) for i in range(B)], chunksize=16)
    res.sort(key=lambda r: r[0])
    print(f"table 1 -- batch {B}: changes = adds + removes after the one-shot purge; (rows of the seed, adds, removes, purged, wrong, missing, final rows)")
    print(f"{'seed':10s} {'rows':>6s} {'mean chg':>9s} {'max chg':>8s} {'p99':>6s} {'mean wrong':>11s} {'max':>4s} {'mean missing':>13s} {'max':>4s}   instance 1613 | instance 1871")
    for name in VARIANTS:
        a = np.array([r[1][name] for r in res])
        chg = a[:, 1] + a[:, 2]
        named = " | ".join(str(tuple(int(x) for x in res[i][1][name])) if i < B else "-" for i in (1613, 1871))
        print(f"{name:10s} {a[:, 0].mean():6.2f} {chg.mean():9.2f} {chg.max():8d} {np.percentile(chg, 99):6.1f} {a[:, 4].mean():11.2f} {a[:, 4].max():4d} "
              f"{a[:, 5].mean():13.2f} {a[:, 5].max():4d}   {named}")
    for c, amp in enumerate(bench.AMPLITUDES):
        idx = np.arange(B) % 3 == c
        print(f"  amplitude {amp}: " + "  ".join(f"{name} {np.array([r[1][name] for r in res])[idx][:, 1:3].sum(1).mean():.2f}/{np.array([r[1][name] for r in res])[idx][:, 1:3].sum(1).max()}" for name in VARIANTS) + "   (mean/max)")
    print("table 2 -- hybrid policies, cycle model of the finish (k cycles), ADMM phase 17 + 5.9 K + 5.6 (+ 5 for the LQR rollout)")
    for name in res[0][2]:
        a = np.array([r[2][name] for r in res])
        tiles = [makespan(a[t * 16:(t + 1) * 16, 1]) for t in range(B // 16)]
        K = int(name[-1])
        admm = 17 + 5.9 * K + 5.6 + (0 if name.startswith("admm") else 5)
        print(f"{name:12s} changes mean {a[:, 0].mean():5.2f} max {int(a[:, 0].max()):3d} zero {int((a[:, 0] == 0).sum()):4d} | slowest instance {a[:, 1].max():6.1f} k | "
              f"tile median {np.median(tiles):5.1f} k max {max(tiles):5.1f} k | step {(admm + max(tiles)):6.1f} k cycles (median tile {(admm + np.median(tiles)):5.1f} k)")
