"""CPU restatement (numpy, float64) of the stage-wise dual active-set solve, csrc/almpc_sdual.hip.h.

TEST INFRASTRUCTURE ONLY (see oracle/mpc_oracle.py): only tests/, __graft_entry__.smoke() and bench.py's checker legs import it.

The QP is the one the reference itself poses in multiple-shooting form (paths relative to /root/reference):
    variables x, u per stage, dynamics as constraints          src/sub/model_modeler_implementation/linear/mpc_modeler_implementation_linear.jl:48-60
    input box                                                  ..linear.jl:73-78
    state box on every stage incl. the fixed first one         ..linear.jl:62-70          (kw mpc_state_constraint)
    terminal equality e_x[:,N+1] = 0                           src/sub/design_mpc.jl:330-331
    cost incl. the input-rate term on delta_u = u_i - u_{i+1}  src/sub/design_mpc.jl:405-468 (S: :423-446)
in deviation coordinates (e_k = x_k - x_ref_k, v_k = u_k - u_ref_k), or the QP of one SQP iteration around a trajectory (stage models
A_k, B_k, defects c_k; oracle/mpc_oracle.py::ltv_qp).

Method: Goldfarb-Idnani dual active set in CONSTRAINT space (as mpc_oracle.solve_qp_dual_active_set), but matrix-free: every row
of the problem -- an input bound, a state bound, a terminal-equality row -- is one coordinate of the stage-wise trajectory
z = (v_0, e_1, v_1, ..., e_N), and a column of  Ghat = A H^-1 A'  is the response of the UNCONSTRAINED problem to a unit linear cost
on that coordinate: one affine backward / forward sweep with the unconstrained (LQR) feedback gains K_k.  The gains do not depend on
the working set, so the Riccati recursion runs once per model (design time for a shared model), never per working-set change, and
the cost of a change is two O(N (n^2 + n m)) sweeps, independent of the spectral radius of A (the forward sweep runs in closed-loop
form).  The input-rate weight S couples neighbouring inputs: the stage state is augmented by the previous input (nt = n + m).
"""
from __future__ import annotations

import numpy as np

__all__ = ["StageQP", "stage_qp_from_problem", "stage_qp_from_ltv", "stage_gains", "stage_sweep", "solve_stage_dual", "solve_mpc_stagewise"]


class StageQP:
    """min  sum_{k=1..N-1} (e_k + ebar_{k-1})'Q(e_k + ebar_{k-1}) + (e_N + ebar_{N-1})'P(e_N + ebar_{N-1})
            + sum_{k=0..N-1} [v_k'R v_k + 2 qu_k'v_k] + sum_{k=0..N-2} (v_k - v_{k+1} + dU_k)'S(v_k - v_{k+1} + dU_k)
       s.t. e_{k+1} = A_k e_k + B_k v_k + c_k, e_0 given; ulo_k <= v_k <= uhi_k; xlo_k <= e_k <= xhi_k (k = 0..N; k = 0 only decides
            feasibility); e_N = eqt (terminal equality)."""

    def __init__(self, A, B, Q, R, S, P, e0, ulo, uhi, c=None, ebar=None, qu=None, dU=None, xlo=None, xhi=None, eqt=None):
        N, m = ulo.shape
        n = e0.size
        self.n, self.m, self.N = n, m, N
        self.A = [np.asarray(A, dtype=np.float64)] * N if np.ndim(A) == 2 else [np.asarray(a, dtype=np.float64) for a in A]
        self.B = [np.asarray(B, dtype=np.float64)] * N if np.ndim(B) == 2 else [np.asarray(b, dtype=np.float64) for b in B]
        self.Q, self.R, self.P = 0.5 * (Q + Q.T), 0.5 * (R + R.T), 0.5 * (P + P.T)
        self.S = np.zeros((m, m)) if S is None else 0.5 * (S + S.T)
        self.useS = bool(np.any(self.S != 0.0))
        self.nt = n + m if self.useS else n
        self.e0 = np.asarray(e0, dtype=np.float64)
        self.ulo, self.uhi = np.asarray(ulo, dtype=np.float64), np.asarray(uhi, dtype=np.float64)
        self.c = np.zeros((N, n)) if c is None else np.asarray(c, dtype=np.float64)
        self.has_c = c is not None
        self.ebar = np.zeros((N, n)) if ebar is None else np.asarray(ebar, dtype=np.float64)
        self.qu = np.zeros((N, m)) if qu is None else np.asarray(qu, dtype=np.float64)
        self.dU = np.zeros((max(N - 1, 0), m)) if dU is None else np.asarray(dU, dtype=np.float64)
        self.xlo = None if xlo is None else np.asarray(xlo, dtype=np.float64)   # (N+1, n)
        self.xhi = None if xhi is None else np.asarray(xhi, dtype=np.float64)
        self.eqt = None if eqt is None else np.asarray(eqt, dtype=np.float64)

    # augmented stage matrices
    def At(self, k):
        if not self.useS:
            return self.A[k]
        n, m = self.n, self.m
        M = np.zeros((n + m, n + m)); M[:n, :n] = self.A[k]
        return M

    def Bt(self, k):
        if not self.useS:
            return self.B[k]
        return np.vstack([self.B[k], np.eye(self.m)])

    def ct(self, k):
        return self.c[k] if not self.useS else np.concatenate([self.c[k], np.zeros(self.m)])


def stage_qp_from_problem(p, x0):
    """The reference's MPC problem (mpc_oracle.MPCProblem) in stage-wise deviation form."""
    useR = p.R[0, 0] != 0.0                       # branch rules of src/sub/design_mpc.jl:423-466
    R = p.R if useR else 0.0 * p.R
    S = p.S if (useR and p.S[0, 0] != 0.0) else None
    e0 = np.asarray(x0, dtype=np.float64) - p.x_ref[:, 0]
    ulo = (p.u_min[:, None] - p.u_ref).T
    uhi = (p.u_max[:, None] - p.u_ref).T
    dU = (p.u_ref[:, :-1] - p.u_ref[:, 1:]).T if S is not None else None   # the rate term is on u = v + u_ref
    xlo = xhi = None
    if p.x_min is not None:
        xlo = (p.x_min[:, None] - p.x_ref).T
        xhi = (p.x_max[:, None] - p.x_ref).T
    eqt = np.zeros(p.n) if p.terminal == "equality" else None
    return StageQP(p.A, p.B, p.Q, R, S, p.P, e0, ulo, uhi, dU=dU, xlo=xlo, xhi=xhi, eqt=eqt)


def stage_qp_from_ltv(A_list, B_list, c_list, xbar, ubar, x_ref, u_ref, Q, R, S, P, u_min, u_max, x_min=None, x_max=None, terminal="none"):
    """The QP of one SQP iteration (mpc_oracle.ltv_qp + ltv_state_rows) in stage-wise form, variable v = u - ubar, dx_0 = 0."""
    xbar, ubar = np.asarray(xbar, dtype=np.float64), np.asarray(ubar, dtype=np.float64)
    n, N1 = xbar.shape
    N = N1 - 1
    m = ubar.shape[0]
    useR = R[0, 0] != 0.0
    Rz = R if useR else 0.0 * R
    Sz = S if (useR and S is not None and S[0, 0] != 0.0) else None
    xr = np.zeros((n, N + 1)) if x_ref is None else np.asarray(x_ref, dtype=np.float64)
    ur = np.zeros((m, N)) if u_ref is None else np.asarray(u_ref, dtype=np.float64)
    ebar = (xbar[:, 1:] - xr[:, 1:]).T
    qu = (Rz @ (ubar - ur)).T
    dU = (ubar[:, :-1] - ubar[:, 1:]).T if Sz is not None else None
    ulo = (np.asarray(u_min)[:, None] - ubar).T
    uhi = (np.asarray(u_max)[:, None] - ubar).T
    xlo = xhi = None
    if x_min is not None:
        xlo = (np.asarray(x_min)[:, None] - xbar).T
        xhi = (np.asarray(x_max)[:, None] - xbar).T
    eqt = (xr[:, N] - xbar[:, N]) if terminal == "equality" else None
    return StageQP(A_list, B_list, Q, Rz, Sz, P, np.zeros(n), ulo, uhi, c=np.asarray(c_list), ebar=ebar, qu=qu, dU=dU, xlo=xlo, xhi=xhi, eqt=eqt)


def stage_gains(q: StageQP):
    """Backward Riccati recursion of the UNCONSTRAINED problem (cross terms from S): per stage K (m x nt), Li = Lam^-1 (m x m),
    Acl = At - Bt K, Lb = Li Bt', and Pc = P_{k+1} ct_k (the defect's share of the value-function gradient); plus the base linear
    terms qx (N+1, nt), qv (N, m) of the cost."""
    n, m, N, nt = q.n, q.m, q.N, q.nt
    qx = np.zeros((N + 1, nt)); qv = np.zeros((N, m))
    Pn = np.zeros((nt, nt)); Pn[:n, :n] = q.P
    qx[N, :n] = q.P @ q.ebar[N - 1]
    K, Li, Acl, Lb, Pc = [None] * N, [None] * N, [None] * N, [None] * N, np.zeros((N, nt))
    for k in range(N - 1, -1, -1):
        At, Bt = q.At(k), q.Bt(k)
        Qt = np.zeros((nt, nt)); M = np.zeros((nt, m)); Rt = q.R.copy()
        qv[k] = q.qu[k]
        if k >= 1:
            Qt[:n, :n] = q.Q
            qx[k, :n] = q.Q @ q.ebar[k - 1]
            if q.useS:
                Qt[n:, n:] = q.S; M[n:, :] = -q.S; Rt = Rt + q.S
                qx[k, n:] = q.S @ q.dU[k - 1]
                qv[k] = qv[k] - q.S @ q.dU[k - 1]
        Lam = Rt + Bt.T @ Pn @ Bt
        Gm = Bt.T @ Pn @ At + M.T
        Li[k] = np.linalg.inv(0.5 * (Lam + Lam.T))
        K[k] = Li[k] @ Gm
        Acl[k] = At - Bt @ K[k]
        Lb[k] = Li[k] @ Bt.T
        Pc[k] = Pn @ q.ct(k)
        Pnew = Qt + At.T @ Pn @ At - Gm.T @ K[k]
        Pn = 0.5 * (Pnew + Pnew.T)
    return dict(K=K, Li=Li, Acl=Acl, Lb=Lb, Pc=Pc, qx=qx, qv=qv)


def stage_sweep(q: StageQP, g, srcx, srcu, x0t, with_c, kback=None, kfwd=None):
    """Affine sweep: minimiser of  J_quad(z) + 2 sum srcx_k'x_k + 2 sum srcu_k'v_k  over the dynamics from x_0 = x0t (defects only
    with_c).  kback: highest stage carrying a source (the backward pass starts there: p_k = 0 above it); kfwd: last stage the caller needs (forward
    pass stops there).  Returns v (N, m), x (N+1, nt)."""
    N, m, nt = q.N, q.m, q.nt
    kf = N if kfwd is None else kfwd
    kff = np.zeros((N, m))
    p = srcx[N].copy()
    kstart = N - 1 if (with_c or kback is None) else min(N - 1, kback)   # above the highest source p and kff are zero
    for k in range(kstart, -1, -1):
        gg = p + (g["Pc"][k] if with_c else 0.0)
        kff[k] = g["Li"][k] @ srcu[k] + g["Lb"][k] @ gg
        p = srcx[k] - g["K"][k].T @ srcu[k] + g["Acl"][k].T @ gg
    v = np.zeros((N, m)); x = np.zeros((N + 1, nt))
    x[0] = x0t
    for k in range(min(kf, N)):
        v[k] = -g["K"][k] @ x[k] - kff[k]
        x[k + 1] = g["Acl"][k] @ x[k] - q.Bt(k) @ kff[k] + (q.ct(k) if with_c else 0.0)
    return v, x


def solve_stage_dual(q: StageQP, v_guess=None, max_iter=None, tol=1e-9, wcap=None, return_info=False):
    """Dual active set on the stage-wise QP.  Rows: ("u", k, a) for input bounds, ("x", k, i) for state bounds (k = 1..N).
    Decisions (shared with the kernel): most violated row, violation measured in units of the row's bound width (1 when a side is
    infinite), ties -> smallest trajectory index; dependent-row test dp <= 1e-12 Ghat_pp; blocking multiplier = smallest step, ties ->
    smallest working-set position.  Returns dict(v (N, m), e (N+1, n), status 0 / 1 / 3, iters, n_active, lam)."""
    n, m, N, nt = q.n, q.m, q.N, q.nt
    g = stage_gains(q)
    big = 1e300
    # coordinates of the trajectory: index t = k (nt + m) + (i | nt + a): x_k at [k(nt+m), +nt), v_k behind it; x_N last
    stride = nt + m
    T = N * stride + nt

    def unpack(v, x):
        z = np.zeros(T)
        for k in range(N):
            z[k * stride:k * stride + nt] = x[k]
            z[k * stride + nt:(k + 1) * stride] = v[k]
        z[N * stride:] = x[N]
        return z

    lo = np.full(T, -big); hi = np.full(T, big); isrow = np.zeros(T, dtype=bool); eq = np.zeros(T, dtype=bool)
    for k in range(N):
        sl = slice(k * stride + nt, (k + 1) * stride)
        lo[sl], hi[sl], isrow[sl] = q.ulo[k], q.uhi[k], True
    if q.xlo is not None:
        for k in range(1, N + 1):
            sl = slice(k * stride, k * stride + n)
            lo[sl], hi[sl], isrow[sl] = q.xlo[k], q.xhi[k], True
    if q.eqt is not None:
        sl = slice(N * stride, N * stride + n)
        lo[sl], hi[sl], isrow[sl], eq[sl] = q.eqt, q.eqt, True, True
    width = np.where((lo > -big) & (hi < big) & (hi > lo), hi - lo, 1.0)
    stage_of = np.minimum(np.arange(T) // stride, N)
    x0t = np.concatenate([q.e0, np.zeros(nt - n)])
    max_iter = 20 * int(isrow.sum()) + 50 if max_iter is None else max_iter
    wcap = T if wcap is None else wcap
    info = dict(sweeps=0)

    if q.xlo is not None and (np.any(q.e0 < q.xlo[0]) or np.any(q.e0 > q.xhi[0])):   # stage 1 of the reference is x0 itself
        v0, x0_ = stage_sweep(q, g, g["qx"], g["qv"], x0t, True)
        out = dict(v=v0, e=x0_[:, :n], status=3, iters=0, n_active=0, lam=np.zeros(T))
        return (out, info) if return_info else out

    def src_arrays(coef):   # linear cost 2 * coef_t on coordinate t
        sx = np.zeros((N + 1, nt)); su = np.zeros((N, m))
        for t in np.flatnonzero(coef):
            k, r = divmod(int(t), stride)
            if k == N or r < nt:
                sx[k, r] = coef[t]
            else:
                su[k, r - nt] = coef[t]
        return sx, su

    def response(coef, kfwd=None):   # Ghat @ coef, as a trajectory
        nzs = np.flatnonzero(coef)
        if nzs.size == 0:
            return np.zeros(T)
        kb = int(max(stage_of[t] for t in nzs))
        sx, su = src_arrays(-0.5 * coef)
        info["sweeps"] += 1
        return unpack(*stage_sweep(q, g, sx, su, np.zeros(nt), False, kback=kb, kfwd=kfwd))

    def full(lam):   # trajectory for multipliers lam, from scratch
        sx, su = src_arrays(0.5 * lam)
        info["sweeps"] += 1
        return unpack(*stage_sweep(q, g, g["qx"] + sx, g["qv"] + su, x0t, True))

    W, side = [], {}
    Sinv = np.zeros((0, 0))
    lam = np.zeros(T)

    def border(j, sd, zj):
        """append row j (response zj) to the working set; False if it depends on the rows already there"""
        nonlocal Sinv
        c = zj[W]
        u = Sinv @ c if W else np.zeros(0)
        sc = zj[j] - (c @ u if W else 0.0)
        if not sc > 1e-12 * zj[j]:
            return False
        k = len(W)
        Sn = np.empty((k + 1, k + 1))
        Sn[:k, :k] = Sinv + np.outer(u, u) / sc
        Sn[:k, k] = -u / sc
        Sn[k, :k] = -u / sc
        Sn[k, k] = 1.0 / sc
        Sinv = Sn
        W.append(j); side[j] = sd
        return True

    def remove(pos):
        nonlocal Sinv
        keep = [i for i in range(len(W)) if i != pos]
        Sinv = Sinv[np.ix_(keep, keep)] - np.outer(Sinv[keep, pos], Sinv[pos, keep]) / Sinv[pos, pos]
        j = W.pop(pos)
        del side[j]
        lam[j] = 0.0

    def bound(j):
        return hi[j] if side[j] >= 0 else lo[j]

    s0 = full(lam)
    s = s0
    it = 0
    # ---- start: equality rows, then the guess's rows on a bound; multipliers of the start, wrong signs purged
    W0 = [(int(t), 0) for t in np.flatnonzero(eq)]
    if v_guess is not None:
        vg = np.clip(np.asarray(v_guess, dtype=np.float64).reshape(N, m), q.ulo, q.uhi)
        for k in range(N):
            for a in range(m):
                if q.ulo[k, a] >= q.uhi[k, a]:
                    continue
                if vg[k, a] >= q.uhi[k, a]:
                    W0.append((k * stride + nt + a, 1))
                elif vg[k, a] <= q.ulo[k, a]:
                    W0.append((k * stride + nt + a, -1))
    if W0:
        kmax = int(max(stage_of[t] for t, _ in W0))
        for t, sd in W0:
            if len(W) >= wcap:
                break
            e = np.zeros(T); e[t] = 1.0
            border(t, sd, response(e, kfwd=min(N, kmax + 1)))
        while W:
            b = np.array([bound(j) for j in W])
            lw = Sinv @ (s0[W] - b)
            viol = [(-lw[i] if side[j] > 0 else lw[i]) if side[j] != 0 else -np.inf for i, j in enumerate(W)]
            if max(viol) <= 1e-12 * max(1.0, float(np.max(np.abs(lw)))):
                lam[:] = 0.0
                lam[W] = lw
                break
            remove(int(np.argmax(viol)))   # (not counted as an iteration: no sweep)
        s = full(lam)
    status = 1
    refined = 0
    while it < max_iter:
        it += 1
        inW = np.zeros(T, dtype=bool); inW[W] = True
        viol = np.where(isrow & ~inW, np.maximum(s - hi, lo - s) / width, -np.inf)
        p_ = int(np.argmax(viol))
        if viol[p_] <= tol:
            # confirmation from scratch: the rows of the working set on their bounds, nothing else violated
            s = full(lam)
            res = np.array([s[j] - bound(j) for j in W]) if W else np.zeros(0)
            viol = np.where(isrow & ~inW, np.maximum(s - hi, lo - s) / width, -np.inf)
            ok = (not W or np.max(np.abs(res) / width[W]) <= 1e-8) and np.max(viol) <= 1e-8
            if ok:
                status = 0
                break
            if refined >= 3:
                break
            refined += 1
            if W:
                lam[W] = lam[W] + Sinv @ res
                s = full(lam)
            continue
        sd = 1 if s[p_] > hi[p_] else -1
        bp = hi[p_] if sd > 0 else lo[p_]
        if len(W) >= wcap:
            break
        lam_p = 0.0
        ep = np.zeros(T); ep[p_] = 1.0
        kmax = int(max([stage_of[p_]] + [stage_of[j] for j in W]))
        zp = response(ep, kfwd=min(N, kmax + 1))
        infeasible = False
        while True:
            c = zp[W]
            r = Sinv @ c if W else np.zeros(0)
            coef = ep.copy()
            coef[W] = -r
            dvec = response(coef)
            dp = dvec[p_]
            dependent = not dp > 1e-12 * zp[p_]
            tau2 = np.inf if dependent else abs(s[p_] - bp) / dp
            tau1, blk, blk_t = np.inf, -1, T
            for i, j in enumerate(W):
                if side[j] == 0:
                    continue
                ri = r[i] * sd          # lam_j moves by -ri * tau (tau >= 0 the step in |lam_p|)
                if side[j] > 0 and ri > 0:
                    cand = max(lam[j], 0.0) / ri
                elif side[j] < 0 and ri < 0:
                    cand = min(lam[j], 0.0) / ri
                else:
                    continue
                if cand < tau1 or (cand == tau1 and j < blk_t):   # ties -> smallest coordinate
                    tau1, blk, blk_t = cand, i, j
            if dependent and blk < 0:
                infeasible = True
                break
            tau = min(tau1, tau2)
            if not dependent:
                s = s - sd * tau * dvec
            lam[W] = lam[W] - sd * tau * r
            lam_p += sd * tau
            if tau1 < tau2:
                remove(blk)
                it += 1
                if it >= max_iter:
                    break
                continue
            border_ok = border(p_, sd, zp)
            if not border_ok:   # (numerically dependent at the very end of the step: treat as reached)
                pass
            else:
                lam[p_] = lam_p
            s[p_] = bp
            break
        if infeasible:
            status = 3
            break
    z = s
    v = np.stack([z[k * stride + nt:(k + 1) * stride] for k in range(N)])
    e = np.stack([z[k * stride:k * stride + n] for k in range(N + 1)])
    if status == 0:
        v = np.clip(v, q.ulo, q.uhi)
    lam_u = np.stack([lam[k * stride + nt:(k + 1) * stride] for k in range(N)])
    lam_x = np.stack([lam[k * stride:k * stride + n] for k in range(N + 1)])
    out = dict(v=v, e=e, status=status, iters=it, n_active=len(W), lam=lam.copy(), W=list(W), lam_u=lam_u, lam_x=lam_x)
    return (out, info) if return_info else out


def solve_mpc_stagewise(p, x0, u_guess=None, **kw):
    """One MPC step of mpc_oracle.MPCProblem through solve_stage_dual: outputs as mpc_oracle.solve_mpc_exact (+ status, iters)."""
    q = stage_qp_from_problem(p, x0)
    vg = None if u_guess is None else (np.asarray(u_guess, dtype=np.float64) - p.u_ref).T
    r = solve_stage_dual(q, v_guess=vg, **kw)
    e_u = r["v"].T.copy()
    e_x = np.zeros((p.n, p.N + 1))
    e_x[:, 0] = q.e0
    for k in range(p.N):   # outputs: the rollout of the inputs (as calculate! reads x from the same solution)
        e_x[:, k + 1] = p.A @ e_x[:, k] + p.B @ e_u[:, k]
    return dict(u=e_u + p.u_ref, e_u=e_u, x=e_x + p.x_ref, e_x=e_x, status=r["status"], iters=r["iters"], n_active=r["n_active"],
                e_solver=r["e"].T)
