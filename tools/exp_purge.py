"""CPU experiment (numpy): active-set operation counts of the polish for the hard instances of the bench workload,
with and without a one-shot purge of wrong-sign rows of the initial guess."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, mpc_oracle as mo, bench

p = mo.quadrotor()
des = mo.design_shared(p, rho=30.0, sigma=1e-6, rho_profile="stiffness")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
X0 = bench.make_x0(mo, 0, B, None)
G = des["G"]; lo = des["lo"]; hi = des["hi"]; nz = lo.size


def polish(v0, z, y, purge, max_iter=3000):
    W, side = [], {}
    w = np.clip(z, lo, hi)
    for j in range(nz):
        if y[j] < 0 and w[j] <= lo[j]: W.append(j); side[j] = -1
        elif y[j] > 0 and w[j] >= hi[j]: W.append(j); side[j] = +1
    k0 = len(W)
    n_add = n_rem = n_purged = 0
    first = True
    for it in range(max_iter):
        if W:
            b = np.array([hi[j] if side[j] > 0 else lo[j] for j in W])
            lam = np.linalg.solve(G[np.ix_(W, W)], v0[W] - b)
            t = v0 - G[:, W] @ lam; t[W] = b
        else:
            lam = np.zeros(0); t = v0.copy()
        if first and purge and W:
            first = False
            viol = np.array([-lam[i] if side[W[i]] > 0 else lam[i] for i in range(len(W))])
            bad = viol > 1e-12 * max(1.0, float(np.max(np.abs(lam))))
            if purge == "all" and bad.any():
                n_purged = int(bad.sum())
                for i in sorted(np.nonzero(bad)[0], reverse=True): del side[W.pop(i)]
                continue
        first = False
        step = t - w
        free = np.ones(nz, bool); free[W] = False
        with np.errstate(divide="ignore", invalid="ignore"):
            r_hi = np.where(free & (t > hi), (hi - w) / step, np.inf)
            r_lo = np.where(free & (t < lo), (lo - w) / step, np.inf)
        rr = np.minimum(r_hi, r_lo); j = int(np.argmin(rr))
        if rr[j] < 1.0:
            w = w + max(rr[j], 0.0) * step
            if r_hi[j] <= r_lo[j]: w[j] = hi[j]; W.append(j); side[j] = +1
            else: w[j] = lo[j]; W.append(j); side[j] = -1
            n_add += 1; continue
        w = t
        if not W: break
        viol = np.array([-lam[i] if side[W[i]] > 0 else lam[i] for i in range(len(W))])
        i = int(np.argmax(viol))
        if viol[i] <= 1e-12 * max(1.0, float(np.max(np.abs(lam)))): break
        del side[W.pop(i)]; n_rem += 1
    return dict(w=np.clip(w, lo, hi), k0=k0, n_add=n_add, n_rem=n_rem, n_purged=n_purged, k=len(W))


rows = []
for i in range(B):
    e0 = X0[i] - p.x_ref[:, 0]
    fs = des["Fs"] @ e0 + des["fS"]
    r = mo.admm_box(des["Hs"], fs, lo, hi, rho=des["rho_vec"], sigma=des["sigma"], alpha=1.6, max_iter=8, check_every=8,
                    Minv=des["Minv"], unscale=des["d"])
    v0 = -G @ fs
    a = polish(v0, r["z"], r["y"], None)
    if a["n_add"] + a["n_rem"] >= 6:
        b = polish(v0, r["z"], r["y"], "all")
        assert np.max(np.abs(a["w"] - b["w"])) < 1e-7, (i, np.max(np.abs(a["w"] - b["w"])))
        rows.append((i, a["k0"], a["n_add"], a["n_rem"], b["n_purged"], b["n_add"], b["n_rem"]))
rows.sort(key=lambda r: -(r[2] + r[3]))
print("inst  k0  base(add,rem)   purge(purged, add, rem)   est cycles base -> purge (add 3.7k, remove 6k, re-init 0.5k/row+2k)")
for r in rows[:25]:
    i, k0, a, rm, pg, a2, rm2 = r
    cb = 3.7 * a + 6 * rm
    cp = 3.7 * a2 + 6 * rm2 + (0.5 * (k0 - pg) + 2 if pg else 0)
    print(f"{i:5d} {k0:3d}   {a:3d} {rm:3d}        {pg:3d} {a2:3d} {rm2:3d}        {cb:6.1f}k -> {cp:6.1f}k")
