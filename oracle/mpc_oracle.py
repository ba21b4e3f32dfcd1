"""CPU oracle (numpy, float64) for the condensed-QP MPC hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product
path: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` may import it, and only as the checker.

What is restated here (reference = AutomationLabsModelPredictiveControl.jl v0.1.4,
paths relative to /root/reference):

* the QP the reference poses through JuMP
    - variables / dynamics / input box / optional state box / deviation defs:
      src/sub/model_modeler_implementation/linear/mpc_modeler_implementation_linear.jl:48-100
    - objective (no 1/2 factor, stage sum i=1..N with Q and R, P on stage N+1 only,
      optional S on delta_u for i=1..N-1): src/sub/design_mpc.jl:405-468
    - terminal cost P = DARE(A,B,Q,R), terminal "equality": src/sub/design_mpc.jl:298-394
    - weights Q=q*I, R=r*I, S=s*I with defaults 100/0.1/0: src/sub/design_mpc.jl:264-283,
      src/main/main_mpc.jl:87-94
    - constant references over the horizon: src/main/main_mpc.jl:105-117
    - per step: fix x[:,1]=x0, optimise, read back u,e_u,x,e_x: src/main/computation_mpc.jl:17-55
* the solver arithmetic the reference delegates to OSQP (third-party, NOT in
  /root/reference: Julia package OSQP compat "0.8", Project.toml:28 -> libosqp 0.6.x,
  exact version unpinned because no Manifest.toml is committed).  Restated from the
  published algorithm (Stellato et al., "OSQP: an operator splitting solver for
  quadratic programs", Alg. 1) in ``osqp_admm`` (general l<=Az<=u form, used on the
  reference's sparse multiple-shooting statement) and ``admm_box`` (the condensed,
  box-only specialisation the HIP kernels implement).

PARITY PIN STATUS: the reference's tests contain no numeric golden vector for a
QP solution.  The oracle is pinned against what the reference's
tests do hold: the linear-regressor fixture (A,B) and the scenario + loose
assertions of test/computation_mpc_test.jl:981-1054, and the structural constraint
counts 74/78 of test/terminal_ingredient_test.jl:160,317.  Beyond those tolerances
(0.5 on x, 3 on u) the OSQP numerics are "parity unpinned"; truth for the 1e-5 bar
is the exact optimum of the reference's strictly convex QP, certified by a KKT
residual (``kkt_residual``) that does not depend on the method that produced it.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Optional

import numpy as np

__all__ = [
    "MPCProblem", "dare", "condense", "condensed_qp", "solve_box_qp_exact", "kkt_residual",
    "rollout", "solve_mpc_exact", "constraint_rows", "solve_qp_dual_active_set", "kkt_general", "sparse_problem", "osqp_admm", "admm_box", "polish_active_set",
    "design_shared", "solve_mpc_admm_polish",
    "jacobi_scaling", "decode_linear_regressor_fixture", "splitmix_normal",
    "double_integrator", "quadrotor", "qtp_linear_fixture_problem", "quadrotor_x0_batch",
    "FnnModel", "synthetic_fnn", "fnn_linear_problem",
]


# --------------------------------------------------------------------------------------
# problem container
# --------------------------------------------------------------------------------------
@dataclasses.dataclass
class MPCProblem:
    """Design data of one linear MPC controller (one `proceed_controller` call).

    Shapes follow the reference's Julia matrices (column-major there; here the
    same logical indices): A n*n, B n*m, Q n*n, R m*m, S m*m, P n*n,
    x_ref n*(N+1), u_ref m*N, u_min/u_max m, x_min/x_max n (or None: the reference
    adds the state box only when kw `mpc_state_constraint` is present,
    ...linear.jl:62-70), terminal in {"none","equality"}.
    """
    A: np.ndarray
    B: np.ndarray
    N: int
    Q: np.ndarray
    R: np.ndarray
    S: np.ndarray
    P: np.ndarray
    x_ref: np.ndarray
    u_ref: np.ndarray
    u_min: np.ndarray
    u_max: np.ndarray
    x_min: Optional[np.ndarray] = None
    x_max: Optional[np.ndarray] = None
    terminal: str = "none"

    @property
    def n(self) -> int:
        return self.A.shape[0]

    @property
    def m(self) -> int:
        return self.B.shape[1]

    @property
    def nz(self) -> int:
        return self.m * self.N


def make_problem(A, B, N, u_min, u_max, x_ref=None, u_ref=None, q=100.0, r=0.1, s=0.0,
                 x_min=None, x_max=None, terminal="none", P=None) -> MPCProblem:
    """Defaults as the reference: Q=100*I, R=0.1*I, S=0 (src/main/main_mpc.jl:87-94),
    P = DARE (src/sub/design_mpc.jl:327), references broadcast over the horizon
    (src/main/main_mpc.jl:105-117)."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    n, m = B.shape
    Q = q * np.eye(n)
    R = r * np.eye(m)
    S = s * np.eye(m)
    if P is None:
        P = dare(A, B, Q, R)
    xr = np.zeros(n) if x_ref is None else np.asarray(x_ref, dtype=np.float64)
    ur = np.zeros(m) if u_ref is None else np.asarray(u_ref, dtype=np.float64)
    x_ref_h = xr.reshape(n, -1) * np.ones((n, N + 1)) if xr.ndim == 1 else xr
    u_ref_h = ur.reshape(m, -1) * np.ones((m, N)) if ur.ndim == 1 else ur
    return MPCProblem(A, B, N, Q, R, S, P, x_ref_h, u_ref_h,
                      np.asarray(u_min, dtype=np.float64), np.asarray(u_max, dtype=np.float64),
                      None if x_min is None else np.asarray(x_min, dtype=np.float64),
                      None if x_max is None else np.asarray(x_max, dtype=np.float64),
                      terminal)


# --------------------------------------------------------------------------------------
# DARE  (reference: ControlSystems.are(Discrete, A, B, Q, R), src/sub/design_mpc.jl:327)
# --------------------------------------------------------------------------------------
def dare(A, B, Q, R, tol=1e-13, max_iter=200):
    """Stabilising solution of A'PA - P - A'PB (R + B'PB)^-1 B'PA + Q = 0 by the
    structure-preserving doubling algorithm (quadratically convergent; also fine
    for eigenvalues on the unit circle as long as (A,B) is stabilisable and
    (A,Q^1/2) detectable).  scipy.linalg.solve_discrete_are is the independent
    check in tests/test_oracle.py."""
    A = np.asarray(A, dtype=np.float64)
    n = A.shape[0]
    Ak = A.copy()
    Gk = B @ np.linalg.solve(R, B.T)
    Hk = np.asarray(Q, dtype=np.float64).copy()
    I = np.eye(n)
    for _ in range(max_iter):
        W = I + Gk @ Hk
        WinvA = np.linalg.solve(W, Ak)
        WinvG = np.linalg.solve(W, Gk)
        A1 = Ak @ WinvA
        G1 = Gk + Ak @ WinvG @ Ak.T
        H1 = Hk + Ak.T @ Hk @ WinvA
        H1 = 0.5 * (H1 + H1.T)
        G1 = 0.5 * (G1 + G1.T)
        done = np.linalg.norm(H1 - Hk, ord="fro") <= tol * max(1.0, np.linalg.norm(H1, ord="fro"))
        Ak, Gk, Hk = A1, G1, H1
        if done:
            break
    return Hk


def dare_residual(A, B, Q, R, P):
    K = np.linalg.solve(R + B.T @ P @ B, B.T @ P @ A)
    return A.T @ P @ A - P - A.T @ P @ B @ K + Q


# --------------------------------------------------------------------------------------
# condensation  (SURVEY.md section 8a "condensed restatement")
# --------------------------------------------------------------------------------------
def condense(p: MPCProblem):
    """E = vec(e_x[:,2..N+1]) = Phi e0 + Gamma v,  v = vec(e_u).

    Phi = [A; A^2; ...; A^N]  (nN x n); Gamma block (i,j) = A^(i-j) B, i>=j  (nN x nz).
    Qbar = blkdiag(Q x (N-1), P): the reference's stage sum runs over columns
    1..N of e_x with Q (column 1 = e0 is a constant) and column N+1 carries only P
    (src/sub/design_mpc.jl:448-456).  OSQP convention 1/2 v'Hv + f'v, so the
    reference's factor-free objective gives H = 2(Gamma' Qbar Gamma + Rbar [+ D'SbarD]),
    F = 2 Gamma' Qbar Phi, f = F e0.
    """
    n, m, N = p.n, p.m, p.N
    Phi = np.zeros((n * N, n))
    Gam = np.zeros((n * N, m * N))
    Ak = np.eye(n)
    G = []  # G[k] = A^k B
    for k in range(N):
        G.append(Ak @ p.B)
        Ak = Ak @ p.A
        Phi[k * n:(k + 1) * n, :] = Ak
    for i in range(N):
        for j in range(i + 1):
            Gam[i * n:(i + 1) * n, j * m:(j + 1) * m] = G[i - j]
    Qbar = np.zeros((n * N, n * N))
    for k in range(N):
        Qbar[k * n:(k + 1) * n, k * n:(k + 1) * n] = p.P if k == N - 1 else p.Q
    H = Gam.T @ Qbar @ Gam
    if p.R[0, 0] != 0.0:  # the reference tests only element [1,1] (src/sub/design_mpc.jl:436,448)
        H = H + np.kron(np.eye(N), p.R)
    # the S term enters only in the branch `R[1,1] != 0 && S[1,1] != 0` (src/sub/design_mpc.jl:436-446)
    if p.R[0, 0] != 0.0 and p.S[0, 0] != 0.0:  # delta_u[:,i] = u[:,i]-u[:,i+1], i=1..N-1
        D = np.zeros((m * (N - 1), m * N))
        for i in range(N - 1):
            D[i * m:(i + 1) * m, i * m:(i + 1) * m] = np.eye(m)
            D[i * m:(i + 1) * m, (i + 1) * m:(i + 2) * m] = -np.eye(m)
        H = H + D.T @ np.kron(np.eye(N - 1), p.S) @ D
    H = 2.0 * H
    H = 0.5 * (H + H.T)
    F = 2.0 * Gam.T @ Qbar @ Phi
    return Phi, Gam, H, F


def s_rate_gradient(p: MPCProblem):
    """Linear term contributed by the S (input-rate) cost when u_ref varies over the
    horizon: delta_u is defined on u (not e_u), u = v + u_ref, so
    sum ||D(v+u_ref)||_S^2 adds 2 D'Sbar D u_ref to f.  Zero for constant references."""
    m, N = p.m, p.N
    if p.S[0, 0] == 0.0 or p.R[0, 0] == 0.0:
        return np.zeros(m * N)
    D = np.zeros((m * (N - 1), m * N))
    for i in range(N - 1):
        D[i * m:(i + 1) * m, i * m:(i + 1) * m] = np.eye(m)
        D[i * m:(i + 1) * m, (i + 1) * m:(i + 2) * m] = -np.eye(m)
    return 2.0 * D.T @ np.kron(np.eye(N - 1), p.S) @ D @ p.u_ref.T.reshape(-1)


def ltv_qp(A_list, B_list, c_list, xbar, ubar, x_ref, u_ref, Q, R, S, P, u_min, u_max, return_prediction=False):
    """QP of one SQP / multiple-shooting iteration around the trajectory (xbar n x (N+1), ubar m x N), in v = u - ubar
    (almpc_design_ltv; BASELINE.json configs[4] -- the reference has no such path, only the QP is reference-shaped):
        dx_{k+1} = A_k dx_k + B_k v_k + c_k, dx_0 = 0   =>   dX = Gam v + g
        cost of src/sub/design_mpc.jl:405-468 evaluated at (xbar + dx, ubar + v), with the same R / S branch rules and the
        S term on u itself,  bounds u_min <= ubar + v <= u_max.
    Returns H, q, lo, hi of  min 1/2 v'Hv + q'v  (and Gam, g with return_prediction: the state rows of a state box / terminal
    equality are rows of dX = Gam v + g, see ltv_state_rows)."""
    N = len(A_list)
    n, m = np.asarray(B_list[0]).shape
    nz = m * N
    Gam = np.zeros((n * N, nz))
    g = np.zeros(n * N)
    row = np.zeros((n, nz))
    gk = np.zeros(n)
    for k in range(N):
        A, B = np.asarray(A_list[k], dtype=np.float64), np.asarray(B_list[k], dtype=np.float64)
        row = A @ row
        row[:, k * m:(k + 1) * m] = B
        gk = A @ gk + (0.0 if c_list is None else np.asarray(c_list[k], dtype=np.float64))
        Gam[k * n:(k + 1) * n] = row
        g[k * n:(k + 1) * n] = gk
    Qbar = np.zeros((n * N, n * N))
    for k in range(N):
        Qbar[k * n:(k + 1) * n, k * n:(k + 1) * n] = P if k == N - 1 else Q
    xr = np.zeros((n, N + 1)) if x_ref is None else np.asarray(x_ref, dtype=np.float64)
    ur = np.zeros((m, N)) if u_ref is None else np.asarray(u_ref, dtype=np.float64)
    ebar = (np.asarray(xbar, dtype=np.float64)[:, 1:] - xr[:, 1:]).T.reshape(-1)
    ub = np.asarray(ubar, dtype=np.float64).T.reshape(-1)
    H = Gam.T @ Qbar @ Gam
    q = Gam.T @ Qbar @ (g + ebar)
    if R[0, 0] != 0.0:
        H = H + np.kron(np.eye(N), R)
        q = q + np.kron(np.eye(N), R) @ (ub - ur.T.reshape(-1))
        if S[0, 0] != 0.0:
            D = np.zeros((m * (N - 1), m * N))
            for i in range(N - 1):
                D[i * m:(i + 1) * m, i * m:(i + 1) * m] = np.eye(m)
                D[i * m:(i + 1) * m, (i + 1) * m:(i + 2) * m] = -np.eye(m)
            DSD = D.T @ np.kron(np.eye(N - 1), S) @ D
            H = H + DSD
            q = q + DSD @ ub
    H = 2.0 * H
    H = 0.5 * (H + H.T)
    q = 2.0 * q
    lo = (np.asarray(u_min, dtype=np.float64)[:, None] - np.asarray(ubar)).T.reshape(-1)
    hi = (np.asarray(u_max, dtype=np.float64)[:, None] - np.asarray(ubar)).T.reshape(-1)
    if return_prediction:
        return H, q, lo, hi, Gam, g
    return H, q, lo, hi


def ltv_state_rows(Gam, g, xbar, x_ref, x_min, x_max, terminal):
    """State rows of one SQP iteration's QP (almpc_set_state_box / almpc_set_terminal_equality with a time-varying design): the box
    x_min <= xbar_k + dx_k <= x_max for stages 2..N+1 (.../fnn/mpc_modeler_implementation_fnn.jl:146-153; stage 1 is x0) and / or
    x_{N+1} = x_ref_{N+1}, i.e. dx_N = x_ref - xbar there (src/sub/design_mpc.jl:330-331), as rows of dX = Gam v + g.
    Returns C, a0, lo_c, hi_c, eq."""
    xbar = np.asarray(xbar, dtype=np.float64)
    n, N1 = xbar.shape
    N = N1 - 1
    rows, lo, hi, eq = [], [], [], []
    for k in range(N):
        for i in range(n):
            is_eq = terminal == "equality" and k == N - 1
            if not (x_min is not None or is_eq):
                continue
            rows.append(k * n + i)
            eq.append(is_eq)
            if is_eq:
                b = x_ref[i, k + 1] - xbar[i, k + 1]
                lo.append(b); hi.append(b)
            else:
                lo.append(x_min[i] - xbar[i, k + 1]); hi.append(x_max[i] - xbar[i, k + 1])
    rows = np.array(rows, dtype=int)
    return Gam[rows], g[rows], np.array(lo), np.array(hi), np.array(eq, dtype=bool)


def solve_qp_rows_exact(H, f, lo, hi, C, a0, lo_c, hi_c, eq_c):
    """Exact solution of  min 1/2 v'Hv + f'v,  lo <= v <= hi,  lo_c <= C v + a0 <= hi_c  (rows eq_c: equalities) by the dual active
    set in constraint space of the Jacobi-scaled problem, KKT-certified; ValueError if infeasible.  Returns v and the active rows."""
    nz = H.shape[0]
    d = jacobi_scaling(H)
    Hs = H * d[:, None] * d[None, :]
    A = np.vstack([np.eye(nz), C * d[None, :]])
    G = np.linalg.inv(Hs)
    Ghat = A @ G @ A.T
    v0 = -G @ (f * d)
    s0 = A @ v0 + np.concatenate([np.zeros(nz), a0])
    lo_r = np.concatenate([lo / d, lo_c])
    hi_r = np.concatenate([hi / d, hi_c])
    eq = np.concatenate([np.zeros(nz, dtype=bool), eq_c])
    r = solve_qp_dual_active_set(Ghat, s0, lo_r, hi_r, eq)
    if r["status"] == 3:
        raise ValueError("infeasible QP")
    w = r["s"][:nz]
    res = kkt_general(Hs, f * d, A, s0 - A @ v0, lo_r, hi_r, w, r["lam"])
    if r["status"] != 0 or res > 1e-7 * max(1.0, float(np.max(np.abs(f * d)))):
        raise RuntimeError(f"general exact solver did not certify: status {r['status']}, KKT residual {res:g}")
    return np.clip(w * d, lo, hi), [j for j in r["W"]]


def condensed_qp(p: MPCProblem, x0):
    """(H, f, lo, hi) of  min 1/2 v'Hv + f'v  s.t. lo <= v <= hi  for initial state x0."""
    Phi, Gam, H, F = condense(p)
    e0 = np.asarray(x0, dtype=np.float64) - p.x_ref[:, 0]
    f = F @ e0 + s_rate_gradient(p)
    lo = (p.u_min[:, None] - p.u_ref).T.reshape(-1)  # stage-major: v = [e_u[:,1]; e_u[:,2]; ...]
    hi = (p.u_max[:, None] - p.u_ref).T.reshape(-1)
    return H, f, lo, hi


def rollout(p: MPCProblem, x0, v):
    """Outputs of calculate! (src/main/computation_mpc.jl:50-53): u, e_u (m x N), x, e_x (n x (N+1))."""
    n, m, N = p.n, p.m, p.N
    e_u = np.asarray(v).reshape(N, m).T
    e_x = np.zeros((n, N + 1))
    e_x[:, 0] = np.asarray(x0, dtype=np.float64) - p.x_ref[:, 0]
    for k in range(N):
        e_x[:, k + 1] = p.A @ e_x[:, k] + p.B @ e_u[:, k]
    return dict(u=e_u + p.u_ref, e_u=e_u, x=e_x + p.x_ref, e_x=e_x)


# --------------------------------------------------------------------------------------
# exact box-QP solver + method-independent certificate
# --------------------------------------------------------------------------------------
def kkt_residual(H, f, lo, hi, v):
    """|| v - clip(v - (Hv+f), lo, hi) ||_inf : zero iff v is the (unique) minimiser."""
    g = H @ v + f
    return float(np.max(np.abs(v - np.clip(v - g, lo, hi)))) if v.size else 0.0


def _solve_fixed(H, f, lo, hi, act_lo, act_hi):
    v = np.where(act_lo, lo, np.where(act_hi, hi, 0.0))
    free = ~(act_lo | act_hi)
    if free.any():
        rhs = -(f[free] + H[np.ix_(free, ~free)] @ v[~free])
        c = np.linalg.cholesky(H[np.ix_(free, free)])
        v[free] = np.linalg.solve(c.T, np.linalg.solve(c, rhs))
    return v


def solve_box_qp_exact(H, f, lo, hi, max_iter=None, tol=1e-9):
    """Exact minimiser of the strictly convex box QP by the classical primal active-set
    method (Nocedal & Wright, Alg. 16.3, specialised to bounds), run in Jacobi-scaled
    coordinates v = D w (D = diag(H)^-1/2) so that multipliers of differently scaled inputs
    are comparable.  No objective evaluations (they cancel catastrophically at
    cond(H) ~ 1e10): feasibility is kept by a ratio test, optimality is decided by
    multiplier signs on the minimiser of the current face.  The result is accepted only
    with the method-independent KKT certificate in scaled coordinates
    (``kkt_residual`` <= tol * max(1, |f'|_inf))."""
    nz = f.size
    if nz == 0:
        return np.zeros(0)
    d = 1.0 / np.sqrt(np.diag(H))
    Hs = H * d[:, None] * d[None, :]
    fs, los, his = f * d, lo / d, hi / d
    scale = max(1.0, float(np.max(np.abs(fs))))
    none = np.zeros(nz, dtype=bool)
    w = np.clip(_solve_fixed(Hs, fs, los, his, none, none), los, his)
    W_lo, W_hi = w <= los, w >= his
    max_iter = 20 * nz + 50 if max_iter is None else max_iter
    for _ in range(max_iter):
        t_full = _solve_fixed(Hs, fs, los, his, W_lo, W_hi)
        step = t_full - w
        free = ~(W_lo | W_hi)
        # ratio test on the free variables
        with np.errstate(divide="ignore", invalid="ignore"):
            r_hi = np.where(free & (step > 0), (his - w) / step, np.inf)
            r_lo = np.where(free & (step < 0), (los - w) / step, np.inf)
        r = np.minimum(r_hi, r_lo)
        j = int(np.argmin(r))
        if r[j] < 1.0:
            w = w + max(r[j], 0.0) * step
            if r_hi[j] <= r_lo[j]:
                w[j] = his[j]
                W_hi[j] = True
            else:
                w[j] = los[j]
                W_lo[j] = True
            continue
        w = t_full
        g = Hs @ w + fs
        viol = np.where(W_lo, -g, np.where(W_hi, g, -np.inf))  # >0 means wrong multiplier sign
        j = int(np.argmax(viol))
        if viol[j] <= 1e-13 * scale:
            break
        W_lo[j] = False
        W_hi[j] = False
    res = kkt_residual(Hs, fs, los, his, w)
    if not res <= tol * scale:
        raise RuntimeError(f"exact box-QP solver did not certify: scaled KKT residual {res:g}")
    return np.clip(w * d, lo, hi)


def constraint_rows(p: MPCProblem, d=None):
    """General-constraint statement of the condensed QP in the scaled variables w (v = d*w; d=None -> unscaled):
    rows of  A = [I; C],  lo_r <= (A w + a0(e0))_r <= hi_r,  where the first nz rows are the input box and the C rows
    are the STATE rows the reference adds: the state box for stages 2..N+1 when `mpc_state_constraint` is present
    (..linear.jl:62-70; stage 1 is the fixed x0 and only decides feasibility) and/or the terminal equality
    e_x[:,N+1] = 0 (src/sub/design_mpc.jl:330-331).  A state row's value is e_x[i,k] = (Phi e0 + Gamma v)[i,k], so its
    offset a0 = Phi e0 depends on the instance while its bounds (x_min - x_ref, x_max - x_ref; 0,0 for the equality)
    are shared.  Returns dict(C (mc x nz), stage (mc,), state (mc,), lo_c, hi_c, eq (mc,) bool)."""
    n, m, N = p.n, p.m, p.N
    Phi, Gam, _, _ = condense(p)
    d = np.ones(p.nz) if d is None else d
    rows, lo, hi, eq, st, ix = [], [], [], [], [], []
    box = p.x_min is not None
    for k in range(N):  # stage k+2 of the reference = block k of E
        for i in range(n):
            is_eq = p.terminal == "equality" and k == N - 1
            if not (box or is_eq):
                continue
            rows.append(k * n + i)
            st.append(k)
            ix.append(i)
            eq.append(is_eq)
            if is_eq:
                lo.append(0.0)
                hi.append(0.0)
            else:
                lo.append(p.x_min[i] - p.x_ref[i, k + 1])
                hi.append(p.x_max[i] - p.x_ref[i, k + 1])
    rows = np.array(rows, dtype=int)
    return dict(C=Gam[rows] * d[None, :] if rows.size else np.zeros((0, p.nz)), Phi=Phi[rows] if rows.size else np.zeros((0, n)),
                lo_c=np.array(lo), hi_c=np.array(hi), eq=np.array(eq, dtype=bool), stage=np.array(st, dtype=int),
                state=np.array(ix, dtype=int))


def solve_qp_dual_active_set(Ghat, s0, lo, hi, eq=None, W0=None, side0=None, max_iter=None, tol=1e-9):
    """Goldfarb-Idnani dual active-set method in CONSTRAINT space for the strictly convex QP
        min 1/2 w'H'w + f'w   s.t.  lo <= A w + a0 <= hi,
    given only  Ghat = A H'^-1 A'  (R x R)  and  s0 = A v0 + a0  (v0 = -H'^-1 f', the unconstrained minimiser):
    with working set W at bounds b, the row values are  s = s0 - Ghat[:,W] lam,  Ghat[W,W] lam = s0[W] - b.
    Dual feasible throughout (lam >= 0 at upper, <= 0 at lower bounds, free on equality rows), primal feasibility is
    restored one violated row at a time, with partial steps when a multiplier would change sign (that row is dropped).
    Returns dict(s, lam (R,), W, iters, status) with status 0 optimal, 1 iteration cap, 3 infeasible."""
    R = s0.size
    eq = np.zeros(R, dtype=bool) if eq is None else eq
    max_iter = 20 * R + 50 if max_iter is None else max_iter
    gn = np.sqrt(np.maximum(np.diag(Ghat), 1e-300))  # row norms in the H'^-1 metric, for the violation measure
    ign = 1.0 / gn                                   # (the measure multiplies by the reciprocal, as k_polish_gen does: same decisions)
    W, side = [], {}
    Sinv = np.zeros((0, 0))
    lam = np.zeros(R)

    def add(j, sd):
        nonlocal Sinv
        c = Ghat[W, j]
        u = Sinv @ c
        sc = Ghat[j, j] - c @ u
        k = len(W)
        Sn = np.empty((k + 1, k + 1))
        Sn[:k, :k] = Sinv + np.outer(u, u) / sc
        Sn[:k, k] = -u / sc
        Sn[k, :k] = -u / sc
        Sn[k, k] = 1.0 / sc
        Sinv = Sn
        W.append(j)
        side[j] = sd

    def remove(pos):
        nonlocal Sinv
        keep = [i for i in range(len(W)) if i != pos]
        Sinv = Sinv[np.ix_(keep, keep)] - np.outer(Sinv[keep, pos], Sinv[pos, keep]) / Sinv[pos, pos]
        j = W.pop(pos)
        del side[j]
        lam[j] = 0.0

    def face():
        if W:
            b = np.array([hi[j] if side[j] >= 0 else lo[j] for j in W])
            lam[:] = 0.0
            lam[W] = Sinv @ (s0[W] - b)
            return s0 - Ghat[:, W] @ lam[W]
        lam[:] = 0.0
        return s0.copy()

    # equality rows are always in the working set; then the warm-start guess
    for j in np.flatnonzero(eq):
        add(int(j), 0)
    if W0 is not None:
        for j, sd in zip(W0, side0):
            if j not in side:
                add(int(j), int(sd))
    s = face()
    # restore dual feasibility of the guess: drop wrong-sign multipliers (worst first)
    it = 0
    while True:
        viol = [(-lam[j] if side[j] > 0 else lam[j]) if side[j] != 0 else -np.inf for j in W]
        if not W or max(viol) <= 1e-12 * max(1.0, np.max(np.abs(lam))):
            break
        remove(int(np.argmax(viol)))
        s = face()
        it += 1
    status = 1
    while it < max_iter:
        it += 1
        inW = np.zeros(R, dtype=bool)
        inW[W] = True
        viol = np.where(inW, -np.inf, np.maximum(s - hi, lo - s) * ign)
        p_ = int(np.argmax(viol))
        if viol[p_] <= tol:
            status = 0
            break
        sd = 1 if s[p_] > hi[p_] else -1
        bp = hi[p_] if sd > 0 else lo[p_]
        lam_p = 0.0
        while True:  # partial steps until row p can be added (or infeasibility shows)
            c = Ghat[W, p_]
            r = Sinv @ c if W else np.zeros(0)
            dvec = Ghat[:, p_] - (Ghat[:, W] @ r if W else 0.0)
            dp = Ghat[p_, p_] - (c @ r if W else 0.0)
            dependent = dp <= 1e-12 * Ghat[p_, p_]
            tau2 = np.inf * sd if dependent else (s[p_] - bp) / dp
            # multipliers that would change sign on the way: lam_i - r_i tau = 0
            tau1, blk = np.inf, -1
            for i, j in enumerate(W):
                if side[j] == 0 or r[i] == 0.0:
                    continue
                ti = lam[j] / r[i]
                if ti * sd > 0 and abs(ti) < tau1 or (lam[j] == 0.0 and r[i] * sd * side[j] > 0 and tau1 > 0):
                    if lam[j] == 0.0:
                        tau1, blk = 0.0, i
                    else:
                        tau1, blk = abs(ti), i
            if dependent and blk < 0:
                return dict(s=s, lam=lam, W=list(W), iters=it, status=3)
            if tau1 < abs(tau2):  # partial step, drop the blocking row, try again
                tau = sd * tau1
                if not dependent:
                    s = s - tau * dvec
                lam[W] = lam[W] - tau * r
                lam_p += tau
                remove(blk)
                it += 1
                if it >= max_iter:
                    break
                continue
            tau = tau2
            s = s - tau * dvec
            lam[W] = lam[W] - tau * r
            lam_p += tau
            add(p_, sd)
            lam[p_] = lam_p
            s[p_] = bp
            break
    return dict(s=s, lam=lam, W=list(W), iters=it, status=status)


def feasibility_slack(p: MPCProblem, x0):
    """Method-independent feasibility certificate of the state-row problem (state box of ..linear.jl:62-70, terminal equality of
    src/sub/design_mpc.jl:330-331): the smallest t for which inputs inside their box exist with  x_min - t <= x_k <= x_max + t
    (k = 2..N+1; the equality e_x[:,N+1] = 0 held exactly), by linear programming (scipy HiGHS).  t <= 0: feasible (|t| = the margin);
    t > 0: infeasible, by that much.  +inf: not even the terminal equality can be met.  Checker only (tests)."""
    from scipy.optimize import linprog
    Phi, Gam, _, _ = condense(p)
    n, m, N, nz = p.n, p.m, p.N, p.nz
    e0 = np.asarray(x0, dtype=np.float64) - p.x_ref[:, 0]
    free = Phi @ e0
    xr = p.x_ref[:, 1:].T.reshape(-1)
    ur = p.u_ref.T.reshape(-1)
    c = np.zeros(nz + 1); c[-1] = 1.0
    A_ub, b_ub = [], []
    if p.x_min is not None:
        hi = np.tile(p.x_max, N) - xr - free
        lo = np.tile(p.x_min, N) - xr - free
        A_ub = [np.hstack([Gam, -np.ones((n * N, 1))]), np.hstack([-Gam, -np.ones((n * N, 1))])]
        b_ub = [hi, -lo]
    A_eq = b_eq = None
    if p.terminal == "equality":
        A_eq = np.hstack([Gam[-n:], np.zeros((n, 1))]); b_eq = -free[-n:]
    bounds = [(p.u_min[k % m] - ur[k], p.u_max[k % m] - ur[k]) for k in range(nz)] + [(None, None)]
    if not A_ub:
        A_ub, b_ub = [np.zeros((1, nz + 1))], [np.zeros(1)]
        A_ub[0][0, -1] = -1.0
    r = linprog(c, A_ub=np.vstack(A_ub), b_ub=np.concatenate(b_ub), A_eq=A_eq, b_eq=b_eq, bounds=bounds, method="highs")
    if r.status != 0:
        return float("inf")
    t = float(r.x[-1])
    if p.x_min is not None:   # stage 1 is x0 itself
        t = max(t, float(np.max(np.maximum(np.asarray(x0) - p.x_max, p.x_min - np.asarray(x0)))))
    return t


def reachability_screen(p: MPCProblem, x0, tol=1e-7):
    """Restatement of k_state_box_screen (csrc/almpc_sdual.hip.h; round 5): a SOUND, incomplete infeasibility test of the state box
    (.../linear/mpc_modeler_implementation_linear.jl:62-70).  Coordinate i of x_k = x_ref_k + A^k e0 + sum_{j<k} A^(k-1-j) B (u_j - u_ref_j)
    ranges over an interval as the inputs range over their box (no other constraint considered: a superset of what is reachable);
    if that interval misses [x_min_i, x_max_i] no admissible input sequence exists.  Returns the first stage (1..N) at which some
    coordinate is unreachable, or 0.  `feasibility_slack` (the phase-1 LP) is the complete answer it must never contradict."""
    if p.x_min is None:
        return 0
    n, N = p.n, p.N
    G = [p.B]
    for _ in range(1, N):
        G.append(p.A @ G[-1])
    e = np.asarray(x0, dtype=np.float64) - p.x_ref[:, 0]
    width = p.x_max - p.x_min
    t = tol * np.where(np.isfinite(width) & (width > 0), width, 1.0)
    for k in range(1, N + 1):
        e = p.A @ e
        lo = np.zeros(n)
        hi = np.zeros(n)
        for j in range(k):
            c0 = G[k - 1 - j] * (p.u_min - p.u_ref[:, j])[None, :]
            c1 = G[k - 1 - j] * (p.u_max - p.u_ref[:, j])[None, :]
            lo += np.minimum(c0, c1).sum(1)
            hi += np.maximum(c0, c1).sum(1)
        xf = e + p.x_ref[:, k]
        if np.any(xf + lo > p.x_max + t) or np.any(xf + hi < p.x_min - t):
            return k
    return 0


def solve_mpc_exact(p: MPCProblem, x0, return_info=False):
    """One MPC step, exact (KKT-certified).  Box-only problems: primal active set on the condensed QP.  With the
    state box and/or the terminal equality: dual active set in constraint space (`solve_qp_dual_active_set`),
    certified by `kkt_general`.  Raises ValueError if the problem is infeasible (the reference would throw from
    JuMP.value)."""
    if p.x_min is None and p.terminal == "none":
        H, f, lo, hi = condensed_qp(p, x0)
        v = solve_box_qp_exact(H, f, lo, hi)
        out = rollout(p, x0, v)
        out["v"] = v
        return out
    if p.terminal not in ("none", "equality"):
        raise NotImplementedError(f"terminal ingredient {p.terminal!r} is not a QP constraint")
    x0 = np.asarray(x0, dtype=np.float64)
    if p.x_min is not None and (np.any(x0 < p.x_min) or np.any(x0 > p.x_max)):
        raise ValueError("infeasible: x[:,1] = x0 violates the state box (..linear.jl:62-70 constrains stage 1 too)")
    H, f, lo, hi = condensed_qp(p, x0)
    d = jacobi_scaling(H)
    Hs = H * d[:, None] * d[None, :]
    cr = constraint_rows(p, d)
    e0 = x0 - p.x_ref[:, 0]
    A = np.vstack([np.eye(p.nz), cr["C"]])
    G = np.linalg.inv(Hs)
    Ghat = A @ G @ A.T
    v0 = -G @ (f * d)
    s0 = A @ v0 + np.concatenate([np.zeros(p.nz), cr["Phi"] @ e0])
    lo_r = np.concatenate([lo / d, cr["lo_c"]])
    hi_r = np.concatenate([hi / d, cr["hi_c"]])
    eq = np.concatenate([np.zeros(p.nz, dtype=bool), cr["eq"]])
    r = solve_qp_dual_active_set(Ghat, s0, lo_r, hi_r, eq)
    if r["status"] == 3:
        raise ValueError("infeasible QP")
    w = r["s"][:p.nz]
    res = kkt_general(Hs, f * d, A, s0 - A @ v0, lo_r, hi_r, w, r["lam"])
    n_state = int(sum(1 for j in r["W"] if j >= p.nz))
    if r["status"] != 0 or res > 1e-7 * max(1.0, float(np.max(np.abs(f * d)))):
        # The bordered inverse of the condensed dual method loses its digits on instances at the edge of feasibility (cond(Ghat_WW)
        # beyond 1e16).  Second opinion by an independent method -- the stage-wise dual active set, whose Ghat columns come from
        # Riccati sweeps instead of the inverse of the condensed Hessian -- judged by the SAME method-independent certificate
        # (kkt_general on the condensed problem); infeasibility is decided by the phase-1 linear programme.
        import stagewise_oracle as so
        r2 = so.solve_stage_dual(so.stage_qp_from_problem(p, x0))
        if r2["status"] == 3 and feasibility_slack(p, x0) > 1e-9:
            raise ValueError("infeasible QP")
        lam2 = np.concatenate([r2["lam_u"].reshape(-1) * d,
                               np.array([r2["lam_x"][k + 1, i] for k, i in zip(cr["stage"], cr["state"])])])
        w = r2["v"].reshape(-1) / d
        res2 = kkt_general(Hs, f * d, A, s0 - A @ v0, lo_r, hi_r, w, lam2)
        if r2["status"] != 0 or res2 > 1e-7 * max(1.0, float(np.max(np.abs(f * d)))):
            raise RuntimeError(f"general exact solver did not certify: status {r['status']}, KKT residual {res:g}; "
                               f"stage-wise second opinion: status {r2['status']}, KKT residual {res2:g}")
        n_state = int(np.count_nonzero(lam2[p.nz:]))
        r = dict(r, iters=r2["iters"], W=[])
    v = np.clip(w * d, lo, hi)
    out = rollout(p, x0, v)
    out["v"] = v
    if return_info:
        out["info"] = dict(n_active_state=n_state, iters=r["iters"], W=r["W"])
    return out


# --------------------------------------------------------------------------------------
# structured (non-condensed) solve: the multiple-shooting form the reference builds (..linear.jl:48-60) solved stage by stage
# --------------------------------------------------------------------------------------
def riccati_active_set(A, B, Q, R, P, e0, lo, hi, c=None, v_guess=None, max_iter=None, tol=1e-9, return_info=False, ebar=None, qu=None):
    """CPU restatement of k_riccati (csrc/almpc_riccati.hip.h): primal active-set method on
        min  e_N'P e_N + sum_{k=1..N-1} e_k'Q e_k + sum_{k=0..N-1} v_k'R v_k
        s.t. e_{k+1} = A_k e_k + B_k v_k + c_k,  e_0 given,  lo_k <= v_k <= hi_k
    (the reference's QP, src/sub/design_mpc.jl:405-468 on ..linear.jl:48-100, in deviation coordinates; stage 1 of the reference is
    e_0 here), whose equality-constrained subproblems -- inputs of the working set held at their bounds -- are solved by a backward
    Riccati recursion and a forward rollout instead of a condensed Hessian: O(N (n^3 + n^2 m)) per working-set change, no m*N limit
    and no loss of definiteness for open-loop unstable models.  A, B: (n, n), (n, m) or lists of N stage matrices; c: (N, n) or None;
    lo, hi: (N, m).  Multipliers come from the adjoint recursion along the trajectory.
    Start: v_guess (N, m) clipped to the box (rows on a bound enter the working set), or the clipped unconstrained (LQR) solution if
    it clips at most 4 inputs, else the reference input (v = 0): most rows a heavily clipped LQR solution holds are not active at the optimum.
    For the QP of an SQP iteration: ebar (N, n) -- the state cost is on e_{k+1} + ebar_k --, qu (N, m) -- J += 2 qu_k'v_k."""
    N, m = lo.shape
    n = e0.size
    Ak = [np.asarray(A)] * N if np.ndim(A) == 2 else [np.asarray(a) for a in A]
    Bk = [np.asarray(B)] * N if np.ndim(B) == 2 else [np.asarray(b) for b in B]
    ck = np.zeros((N, n)) if c is None else np.asarray(c, dtype=np.float64)
    eb = np.zeros((N, n)) if ebar is None else np.asarray(ebar, dtype=np.float64)
    quk = np.zeros((N, m)) if qu is None else np.asarray(qu, dtype=np.float64)
    Qs, Rs, Ps = 0.5 * (Q + Q.T), 0.5 * (R + R.T), 0.5 * (P + P.T)
    max_iter = 20 * N * m + 50 if max_iter is None else max_iter

    def sweep(fixed, bval):
        """backward Riccati for the working set (fixed (N, m) bool, bval (N, m)), then the forward rollout: v*, e*"""
        Kst, kst = [None] * N, [None] * N
        Pn, pn = Ps.copy(), Ps @ eb[N - 1]
        for k in range(N - 1, -1, -1):
            a, b = Ak[k], Bk[k]
            bt = np.where(fixed[k], bval[k], 0.0)
            d = b @ bt + ck[k]
            g = Pn @ d + pn
            M1 = Pn @ a
            BPA = b.T @ M1
            Lam = Rs + b.T @ (Pn @ b)
            h = b.T @ g + Rs @ bt + quk[k]
            fx = fixed[k]
            Lam = Lam.copy()
            Lam[fx, :] = 0.0; Lam[:, fx] = 0.0; Lam[fx, fx] = 1.0
            BPA = BPA.copy(); BPA[fx, :] = 0.0
            h = h.copy(); h[fx] = 0.0
            Li = np.linalg.inv(Lam)
            K, kff = Li @ BPA, Li @ h
            Kst[k], kst[k] = K, kff
            if k > 0:
                Pnew = Qs + a.T @ M1 - BPA.T @ K
                pn = Qs @ eb[k - 1] + a.T @ g - BPA.T @ kff
                Pn = 0.5 * (Pnew + Pnew.T)
        vs, es = np.empty((N, m)), np.empty((N + 1, n))
        es[0] = e0
        for k in range(N):
            vs[k] = np.where(fixed[k], bval[k], -Kst[k] @ es[k] - kst[k])
            es[k + 1] = Ak[k] @ es[k] + Bk[k] @ vs[k] + ck[k]
        return vs, es

    fixed = np.zeros((N, m), dtype=bool)
    side = np.zeros((N, m), dtype=int)
    bval = np.zeros((N, m))
    if v_guess is None:
        vs, _ = sweep(fixed, bval)
        if int(((vs > hi) | (vs < lo)).sum()) > 4:   # k_riccati's RICCATI_CLIP_START: many clipped rows -> start from the reference input
            vs = np.zeros((N, m))
        v = np.clip(vs, lo, hi)
    else:
        v = np.clip(np.asarray(v_guess, dtype=np.float64).reshape(N, m), lo, hi)
    up, dn = v >= hi, v <= lo
    fixed = up | dn
    side = np.where(up, 1, np.where(dn, -1, 0))
    bval = np.where(up, hi, np.where(dn, lo, 0.0))
    n_add = n_rem = 0
    status = 1
    it = 0
    for it in range(1, max_iter + 1):
        vs, es = sweep(fixed, bval)
        free = ~fixed
        step = vs - v
        with np.errstate(divide="ignore", invalid="ignore"):
            r_hi = np.where(free & (vs > hi), (hi - v) / step, np.inf)
            r_lo = np.where(free & (vs < lo), (lo - v) / step, np.inf)
        rr = np.minimum(r_hi, r_lo)
        j = int(np.argmin(rr))          # ties: smallest stage, then smallest input
        kj, aj = divmod(j, m)
        if rr[kj, aj] < 1.0:
            t = max(rr[kj, aj], 0.0)
            v = np.where(free, v + t * step, v)
            if r_hi[kj, aj] <= r_lo[kj, aj]:
                v[kj, aj] = hi[kj, aj]; side[kj, aj] = 1; bval[kj, aj] = hi[kj, aj]
            else:
                v[kj, aj] = lo[kj, aj]; side[kj, aj] = -1; bval[kj, aj] = lo[kj, aj]
            fixed[kj, aj] = True
            n_add += 1
            continue
        v = vs.copy()
        # multipliers of the bounds from the adjoint recursion: mu_k = 2 (R v_k + B_k' lam_{k+1}), lam_N = P e_N, lam_k = Q e_k + A_k' lam_{k+1}
        lam = Ps @ (es[N] + eb[N - 1])
        mu = np.empty((N, m))
        for k in range(N - 1, -1, -1):
            mu[k] = 2.0 * (Rs @ v[k] + quk[k] + Bk[k].T @ lam)
            lam = (Qs @ (es[k] + eb[k - 1]) if k > 0 else 0.0) + Ak[k].T @ lam
        viol = np.where(fixed, side * mu, -np.inf)      # an upper bound needs mu <= 0, a lower bound mu >= 0
        mmax = float(np.max(np.abs(np.where(fixed, mu, 0.0)))) if fixed.any() else 0.0
        j = int(np.argmax(viol))
        kj, aj = divmod(j, m)
        if not fixed.any() or viol[kj, aj] <= tol * max(1.0, mmax):
            status = 0
            break
        fixed[kj, aj] = False; side[kj, aj] = 0; bval[kj, aj] = 0.0
        n_rem += 1
    vs, es = sweep(fixed, bval)
    v = np.clip(vs, lo, hi) if status == 0 else v
    es = np.empty((N + 1, n)); es[0] = e0
    for k in range(N):
        es[k + 1] = Ak[k] @ es[k] + Bk[k] @ v[k] + ck[k]
    out = dict(v=v, e=es, status=status, iters=it, n_add=n_add, n_remove=n_rem, n_active=int(fixed.sum()))
    return out


def solve_mpc_structured(p: MPCProblem, x0, u_guess=None, **kw):
    """One MPC step through `riccati_active_set` (box-only problems, S = 0): same outputs as `solve_mpc_exact`."""
    if p.x_min is not None or p.terminal != "none":
        raise NotImplementedError("structured solve: input box only")
    if p.R[0, 0] != 0.0 and p.S[0, 0] != 0.0:
        raise NotImplementedError("structured solve: no input-rate weight")
    R = p.R if p.R[0, 0] != 0.0 else 0.0 * p.R          # the reference's branch rule (src/sub/design_mpc.jl:423-466)
    e0 = np.asarray(x0, dtype=np.float64) - p.x_ref[:, 0]
    lo = (p.u_min[:, None] - p.u_ref).T
    hi = (p.u_max[:, None] - p.u_ref).T
    vg = None if u_guess is None else (np.asarray(u_guess, dtype=np.float64) - p.u_ref).T
    r = riccati_active_set(p.A, p.B, p.Q, R, p.P, e0, lo, hi, v_guess=vg, **kw)
    e_u = r["v"].T
    e_x = r["e"].T
    return dict(u=e_u + p.u_ref, e_u=e_u, x=e_x + p.x_ref, e_x=e_x, status=r["status"], iters=r["iters"], n_active=r["n_active"],
                n_add=r["n_add"], n_remove=r["n_remove"])


def kkt_general(Hs, fs, A, a0, lo, hi, w, lam):
    """Method-independent optimality certificate for  min 1/2 w'Hs w + fs'w  s.t. lo <= A w + a0 <= hi :
    max of stationarity |Hs w + fs + A'lam|_inf, primal violation, and complementarity/sign violation of lam."""
    s = A @ w + a0
    stat = np.max(np.abs(Hs @ w + fs + A.T @ lam))
    prim = max(0.0, float(np.max(s - hi)), float(np.max(lo - s)))
    at_hi = np.isclose(s, hi, rtol=0, atol=1e-9 * (1 + np.abs(hi)))
    at_lo = np.isclose(s, lo, rtol=0, atol=1e-9 * (1 + np.abs(lo)))
    sign = 0.0
    for i in range(lam.size):
        if lam[i] > 0 and not at_hi[i]:
            sign = max(sign, lam[i])
        if lam[i] < 0 and not at_lo[i]:
            sign = max(sign, -lam[i])
    return float(max(stat, prim, sign))


# --------------------------------------------------------------------------------------
# the reference's own (sparse, multiple-shooting) statement of the QP
# --------------------------------------------------------------------------------------
def sparse_problem(p: MPCProblem, x0):
    """The QP exactly as the reference builds it in JuMP, in OSQP form
    min 1/2 z'Pz + q'z  s.t.  l <= Az <= u,  z = [x; e_x; x_reference; u; e_u; u_reference]
    (variable blocks in declaration order, ...linear.jl:48-55; each block column-major).

    Rows, in the order the reference adds them: dynamics (..linear.jl:58-60), optional
    state box (:62-70), input box (:73-78, two one-sided rows per input and stage),
    e_x defs (:81-83), e_u defs (:85-87), optional terminal equality
    (src/sub/design_mpc.jl:330-331); then the `JuMP.fix`ed variables as equality rows
    (x_reference, u_reference :90-100; x[:,1] src/main/computation_mpc.jl:23-27).

    Returns dict(P,q,A,l,u, idx=block offsets, n_constraints=<number of JuMP constraint
    rows, i.e. excluding fixed-variable bounds> ) -- n_constraints is what
    `JuMP.num_constraints(...; count_variable_in_set_constraints=false)` would count and
    is pinned to 74 / 78 by test/terminal_ingredient_test.jl:160,317 for N=5,n=4,m=2.
    """
    n, m, N = p.n, p.m, p.N
    nx, nu = n * (N + 1), m * N
    off = {}
    o = 0
    for name, sz in (("x", nx), ("e_x", nx), ("x_reference", nx), ("u", nu), ("e_u", nu), ("u_reference", nu)):
        off[name] = o
        o += sz
    nv = o
    ix = lambda name, i, k: off[name] + i + (n if name in ("x", "e_x", "x_reference") else m) * k
    rows, lo, hi = [], [], []

    def add(coefs, l, u):
        r = np.zeros(nv)
        for j, c in coefs:
            r[j] += c
        rows.append(r)
        lo.append(l)
        hi.append(u)

    for k in range(N):  # e_x[:,k+1] == A e_x[:,k] + B e_u[:,k]
        for i in range(n):
            co = [(ix("e_x", i, k + 1), 1.0)]
            co += [(ix("e_x", j, k), -p.A[i, j]) for j in range(n)]
            co += [(ix("e_u", j, k), -p.B[i, j]) for j in range(m)]
            add(co, 0.0, 0.0)
    if p.x_min is not None:
        for k in range(N + 1):
            for i in range(n):
                add([(ix("x", i, k), 1.0)], -np.inf, p.x_max[i])
                add([(ix("x", i, k), 1.0)], p.x_min[i], np.inf)
    for k in range(N):
        for i in range(m):
            add([(ix("u", i, k), 1.0)], -np.inf, p.u_max[i])
            add([(ix("u", i, k), 1.0)], p.u_min[i], np.inf)
    for k in range(N + 1):
        for i in range(n):
            add([(ix("e_x", i, k), 1.0), (ix("x", i, k), -1.0), (ix("x_reference", i, k), 1.0)], 0.0, 0.0)
    for k in range(N):
        for i in range(m):
            add([(ix("e_u", i, k), 1.0), (ix("u", i, k), -1.0), (ix("u_reference", i, k), 1.0)], 0.0, 0.0)
    if p.terminal == "equality":
        for i in range(n):
            add([(ix("e_x", i, N), 1.0)], 0.0, 0.0)
    n_constraints = len(rows)
    if p.S[0, 0] != 0.0:
        raise NotImplementedError("delta_u variables of the S branch are not part of this sparse statement")
    # fixed variables
    for k in range(N + 1):
        for i in range(n):
            add([(ix("x_reference", i, k), 1.0)], p.x_ref[i, k], p.x_ref[i, k])
    for k in range(N):
        for i in range(m):
            add([(ix("u_reference", i, k), 1.0)], p.u_ref[i, k], p.u_ref[i, k])
    for i in range(n):
        add([(ix("x", i, 0), 1.0)], float(x0[i]), float(x0[i]))
    Pm = np.zeros((nv, nv))
    for k in range(N):  # stages 1..N: Q on e_x[:,k], R on e_u[:,k]  (objective has no 1/2 -> P = 2*blk)
        s = off["e_x"] + n * k
        Pm[s:s + n, s:s + n] += 2.0 * p.Q
        if p.R[0, 0] != 0.0:
            s = off["e_u"] + m * k
            Pm[s:s + m, s:s + m] += 2.0 * p.R
    s = off["e_x"] + n * N
    Pm[s:s + n, s:s + n] += 2.0 * p.P
    return dict(P=Pm, q=np.zeros(nv), A=np.array(rows), l=np.array(lo), u=np.array(hi), idx=off,
                n_constraints=n_constraints, nv=nv)


# --------------------------------------------------------------------------------------
# OSQP iteration restated (general form) -- third-party arithmetic, see module docstring
# --------------------------------------------------------------------------------------
def osqp_admm(P, q, A, l, u, rho=0.1, sigma=1e-6, alpha=1.6, eps_abs=1e-3, eps_rel=1e-3,
              max_iter=4000, check_every=25, adaptive_rho=True, x0=None, y0=None):
    """OSQP Algorithm 1 with per-row rho (1e3*rho on equality rows), termination every
    `check_every` iterations on unscaled residuals, adaptive rho with refactorisation when
    the estimate moves by more than 5x.  Dense linear algebra (oracle sizes only).  No Ruiz
    scaling (it changes the iterates, not the fixed point)."""
    nv, nc = P.shape[0], A.shape[0]
    eq = (l == u)
    x = np.zeros(nv) if x0 is None else x0.copy()
    y = np.zeros(nc) if y0 is None else y0.copy()
    z = np.clip(A @ x, l, u)

    def factor(rho_):
        rv = np.where(eq, 1e3 * rho_, rho_)
        K = np.block([[P + sigma * np.eye(nv), A.T], [A, -np.diag(1.0 / rv)]])
        import scipy.linalg as sla
        return rv, sla.lu_factor(K)

    import scipy.linalg as sla
    rv, lu = factor(rho)
    status, it = 1, 0
    for it in range(1, max_iter + 1):
        rhs = np.concatenate([sigma * x - q, z - y / rv])
        sol = sla.lu_solve(lu, rhs)
        xt, nu = sol[:nv], sol[nv:]
        zt = z + (nu - y) / rv
        x = alpha * xt + (1 - alpha) * x
        w = alpha * zt + (1 - alpha) * z
        zn = np.clip(w + y / rv, l, u)
        y = y + rv * (w - zn)
        z = zn
        if it % check_every == 0 or it == max_iter:
            Ax, Px, Aty = A @ x, P @ x, A.T @ y
            rp = np.max(np.abs(Ax - z))
            rd = np.max(np.abs(Px + q + Aty))
            np_ = max(np.max(np.abs(Ax)), np.max(np.abs(z)))
            nd_ = max(np.max(np.abs(Px)), np.max(np.abs(Aty)), np.max(np.abs(q)))
            if rp <= eps_abs + eps_rel * np_ and rd <= eps_abs + eps_rel * nd_:
                status = 0
                break
            if adaptive_rho:
                est = rho * math.sqrt((rp / max(np_, 1e-300)) / max(rd / max(nd_, 1e-300), 1e-300))
                est = min(max(est, 1e-6), 1e6)
                if est > 5 * rho or est < rho / 5:
                    rho = est
                    rv, lu = factor(rho)
    return dict(x=x, z=z, y=y, iters=it, status=status, rho=rho)


# --------------------------------------------------------------------------------------
# condensed, box-only specialisation: the algorithm the HIP kernels implement
# --------------------------------------------------------------------------------------
def jacobi_scaling(H):
    """Diagonal scaling v = D w with D = diag(H)^-1/2: H' = D H D has unit diagonal and the
    box stays a box (lo/D <= w <= hi/D).  This is the k=1 pass of OSQP's Ruiz equilibration
    restricted to the variables block."""
    return 1.0 / np.sqrt(np.diag(H))


def admm_box(Hs, fs, lo, hi, rho=0.1, sigma=1e-6, alpha=1.6, eps_abs=1e-3, eps_rel=1e-3,
             max_iter=4000, check_every=25, Minv=None, unscale=None, x0=None, y0=None):
    """OSQP iteration for constraint matrix A = I (box form) on an
    already scaled problem (Hs, fs, lo, hi in the w coordinates of ``design_shared``);
    `unscale` = d maps back (v = d*w) so that residuals and the termination test are
    evaluated on the UNSCALED problem, as OSQP does with its own scaling.

        xt   = (Hs + (sigma+rho) I)^-1 (sigma x - fs + rho z - y)
        x+   = alpha xt + (1-alpha) x
        w    = alpha xt + (1-alpha) z + y/rho
        z+   = clip(w, lo, hi);   y+ = rho (w - z+)

    Termination (every `check_every` iterations, and at max_iter):
        |d(x-z)|_inf <= eps_abs + eps_rel max(|dx|_inf, |dz|_inf)   and
        |(Hs x + fs + y)/d|_inf <= eps_abs + eps_rel max(|Hs x/d|_inf, |y/d|_inf, |fs/d|_inf).
    Returns x, z, y (scaled coordinates), iters, status (0 solved, 1 max-iter)."""
    nz = fs.size
    d = np.ones(nz) if unscale is None else unscale
    rho = np.full(nz, float(rho)) if np.ndim(rho) == 0 else np.asarray(rho, dtype=np.float64)  # per-row penalty
    if Minv is None:
        Minv = np.linalg.inv(Hs + sigma * np.eye(nz) + np.diag(rho))
    x = np.zeros(nz) if x0 is None else x0.copy()
    y = np.zeros(nz) if y0 is None else y0.copy()
    z = np.clip(x, lo, hi)
    status, it = 1, 0
    for it in range(1, max_iter + 1):
        xt = Minv @ (sigma * x - fs + rho * z - y)
        x = alpha * xt + (1 - alpha) * x
        w = alpha * xt + (1 - alpha) * z + y / rho
        zn = np.clip(w, lo, hi)
        y = rho * (w - zn)
        z = zn
        if it % check_every == 0 or it == max_iter:
            Hx = Hs @ x
            rp = np.max(np.abs(d * (x - z)))
            rd = np.max(np.abs((Hx + fs + y) / d))
            np_ = max(np.max(np.abs(d * x)), np.max(np.abs(d * z)))
            nd_ = max(np.max(np.abs(Hx / d)), np.max(np.abs(y / d)), np.max(np.abs(fs / d)))
            if rp <= eps_abs + eps_rel * np_ and rd <= eps_abs + eps_rel * nd_:
                status = 0
                break
    return dict(x=x, z=z, y=y, iters=it, status=status)


def saturated_lqr_rollout(p: "MPCProblem", x0):
    """Seed candidate for the active-set finish (round-4 review, item 1; measured in tools/exp_lqr_seed.py and NOT adopted,
    DESIGN.md section 4): with P = DARE (src/sub/design_mpc.jl:327) the unconstrained optimum of the reference's QP is the LQR
    feedback, so roll the plant forward under u_k = clip(-K e_k) with the clipping propagated through the state,
    e_{k+1} = A e_k + B u_k.  Returns (v, side): the feasible input sequence v = vec(e_u) (unscaled, stage-major like `condense`)
    and side[j] in {-1, 0, +1}: row j was cut off at its lower / upper bound."""
    K = np.linalg.solve(p.R + p.B.T @ p.P @ p.B, p.B.T @ p.P @ p.A)
    e = np.asarray(x0, dtype=np.float64) - p.x_ref[:, 0]
    v = np.empty((p.N, p.m))
    side = np.zeros((p.N, p.m), dtype=int)
    for k in range(p.N):
        lo_k, hi_k = p.u_min - p.u_ref[:, k], p.u_max - p.u_ref[:, k]
        uk = -K @ e
        side[k] = np.where(uk < lo_k, -1, np.where(uk > hi_k, 1, 0))
        v[k] = np.clip(uk, lo_k, hi_k)
        e = p.A @ e + p.B @ v[k]
    return v.reshape(-1), side.reshape(-1)


def polish_active_set(G, v0, lo, hi, z, y, max_iter=None, refine=True, seed=None):
    """Polish = exact active-set finish started from the ADMM iterate, in the scaled
    coordinates of ``design_shared`` (OSQP's polish guesses the active rows from the sign of
    y and solves one equality-constrained KKT system; here the guess is then corrected by
    the primal active-set method until the multipliers have the right signs, which is what
    makes the 1e-5 bar reachable at cond(H) ~ 1e10).

    Only the SHARED inverse G = H'^-1 is used.  With working set W held at its bounds b
        t = v0 - G[:,W] lam,     G[W,W] lam = v0[W] - b,     v0 = -G f'
    is the minimiser of the face, and lam are the bound multipliers (H't + f' + E_W lam = 0),
    so a sweep needs a |W|x|W| solve instead of an nz x nz one.  (G[W,W])^-1 is kept
    explicitly and updated by bordering (add) / Schur down-dating (remove), the form the
    HIP kernel uses because every step of it is lane-parallel.

    `seed` (optional, one entry per row: -1 / 0 / +1) replaces the sign-of-y guess: those rows start in the working set at
    their lower / upper bound (z must be feasible; it is moved onto the bounds of the seeded rows).  The optimum does not
    depend on the seed, only the number of changes does (tools/exp_lqr_seed.py).

    Returns dict(w, iters, n_add, n_remove, n_active, n_purged, side).
    """
    nz = v0.size
    max_iter = 2 * nz + 50 if max_iter is None else max_iter
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

    w = np.clip(z, lo, hi)
    for j in range(nz):
        if seed is not None:
            if seed[j] != 0:
                w[j] = lo[j] if seed[j] < 0 else hi[j]
                add(j, -1 if seed[j] < 0 else +1)
        elif y[j] < 0 and w[j] <= lo[j]:
            add(j, -1)
        elif y[j] > 0 and w[j] >= hi[j]:
            add(j, +1)
    n_add = n_rem = n_purged = 0
    # One-shot purge (as in k_polish and almpc_oracle.c): two or more wrong-sign multipliers on a guessed set of at
    # most 32 rows -> drop those rows together and rebuild the inverse for the others.
    if 0 < len(W) <= 32:
        b = np.array([hi[j] if side[j] > 0 else lo[j] for j in W])
        lam = Sinv @ (v0[W] - b)
        viol = np.array([-lam[i] if side[W[i]] > 0 else lam[i] for i in range(len(W))])
        bad = viol > 1e-12 * max(1.0, float(np.max(np.abs(lam))))
        if bad.sum() >= 2:
            kept = [(W[i], side[W[i]]) for i in range(len(W)) if not bad[i]]
            n_rem += int(bad.sum())
            n_purged = int(bad.sum())
            W, side, Sinv = [], {}, np.zeros((0, 0))
            for j, sd in kept:
                add(j, sd)
    it = 0
    for it in range(1, max_iter + 1):
        if W:
            b = np.array([hi[j] if side[j] > 0 else lo[j] for j in W])
            r = v0[W] - b
            lam = Sinv @ r
            if refine:
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
    return dict(w=np.clip(w, lo, hi), iters=it + n_purged, n_add=n_add, n_remove=n_rem, n_active=len(W), n_purged=n_purged, side=fin)


def design_shared(p: MPCProblem, rho=0.1, sigma=1e-6, rho_profile="scalar"):
    """Everything `almpc_design_shared` precomputes for a model shared by the whole batch,
    in scaled coordinates v = D w:  d, H' = DHD, F' = DF, Minv = (H' + (sigma+rho) I)^-1,
    G = H'^-1, lo' = lo/d, hi' = hi/d, plus the unscaled H, F (for `almpc_get_design`)."""
    Phi, Gam, H, F = condense(p)
    d = jacobi_scaling(H)
    Hs = H * d[:, None] * d[None, :]
    Hs = 0.5 * (Hs + Hs.T)
    nz = p.nz
    lo = (p.u_min[:, None] - p.u_ref).T.reshape(-1)
    hi = (p.u_max[:, None] - p.u_ref).T.reshape(-1)

    def spd_inv(M):
        c = np.linalg.cholesky(M)
        ci = np.linalg.solve(c, np.eye(nz))
        return ci.T @ ci

    G = spd_inv(Hs)
    # ADMM penalty per row: OSQP's scalar rho, or the stiffness profile rho / G_ii (almpc_set_rho_profile)
    rho_vec = rho / np.diag(G) if rho_profile == "stiffness" else np.full(nz, float(rho))
    return dict(H=H, F=F, d=d, Hs=Hs, Fs=F * d[:, None], fS=s_rate_gradient(p) * d,
                Minv=spd_inv(Hs + sigma * np.eye(nz) + np.diag(rho_vec)), G=G,
                lo=lo / d, hi=hi / d, rho=rho, rho_vec=rho_vec, sigma=sigma, Phi=Phi, Gam=Gam)


def solve_mpc_admm_polish(p: MPCProblem, x0, des=None, rho=0.1, sigma=1e-6, alpha=1.6,
                          eps_abs=1e-3, eps_rel=1e-3, max_iter=4000, check_every=25, polish=True):
    """One MPC step by the algorithm of the HIP path: scaled box ADMM to OSQP's default
    tolerance, then the active-set polish.  Returns rollout dict + v, iters, status,
    polish_iters."""
    des = design_shared(p, rho, sigma) if des is None else des
    d = des["d"]
    e0 = np.asarray(x0, dtype=np.float64) - p.x_ref[:, 0]
    fs = des["Fs"] @ e0 + des["fS"]
    r = admm_box(des["Hs"], fs, des["lo"], des["hi"], rho=des["rho_vec"], sigma=des["sigma"], alpha=alpha,
                 eps_abs=eps_abs, eps_rel=eps_rel, max_iter=max_iter, check_every=check_every,
                 Minv=des["Minv"], unscale=d)
    w, pit = r["z"], 0
    if polish:
        pol = polish_active_set(des["G"], -des["G"] @ fs, des["lo"], des["hi"], r["z"], r["y"])
        w, pit = pol["w"], pol["iters"]
    v = w * d
    out = rollout(p, x0, v)
    out.update(v=v, iters=r["iters"], status=r["status"], polish_iters=pit)
    return out


# --------------------------------------------------------------------------------------
# black-box (Fnn) models: linearise, then the linear path   (reference: .../fnn/mpc_modeler_implementation_fnn.jl:23-58)
# --------------------------------------------------------------------------------------
@dataclasses.dataclass
class FnnModel:
    """Discrete-time model x+ = f(x,u) in the layout the reference reads from Flux.params
    (.../fnn/mpc_modeler_implementation_fnn.jl:88-107, 127-144):
        y1 = W_in [x;u]                      (no bias, no activation)
        yj = act(W_h[j-2] y(j-1) + b_h[j-2]) j = 2..L+1
        x+ = W_out y(L+1)                    (no bias)
    act in {"identity", "relu", "tanh", "sigmoid", "swish"} (NNlib names; the activation found at f[2][1].sigma,
    src/sub/design_mpc.jl:472-483; the NLP branch registers it as a JuMP function, .../fnn/...:120-122)."""
    W_in: np.ndarray          # H x (n+m)
    W_h: list                 # L matrices H x H
    b_h: list                 # L vectors H
    W_out: np.ndarray         # n x H
    act: str = "relu"

    def _act(self, a):
        """activation value and derivative"""
        if self.act == "relu":
            return np.maximum(a, 0.0), (a > 0).astype(np.float64)
        if self.act == "tanh":
            t = np.tanh(a)
            return t, 1.0 - t * t
        if self.act in ("sigmoid", "swish"):
            s = 1.0 / (1.0 + np.exp(-a))
            return (s, s * (1.0 - s)) if self.act == "sigmoid" else (a * s, s * (1.0 + a * (1.0 - s)))
        if self.act == "identity":
            return a, np.ones_like(a)
        raise ValueError(self.act)

    def forward(self, x, u):
        y = self.W_in @ np.concatenate([x, u])
        for W, b in zip(self.W_h, self.b_h):
            y = self._act(W @ y + b)[0]
        return self.W_out @ y

    def jacobian(self, x, u):
        """(A, B) = d f / d x, d f / d u at (x, u): what AutomationLabsSystems.proceed_system_linearization returns
        as .A, .B (the package is not in /root/reference; the reference uses only these two fields and ignores the
        affine offset)."""
        y = self.W_in @ np.concatenate([x, u])
        J = self.W_in.copy()
        for W, b in zip(self.W_h, self.b_h):
            y, der = self._act(W @ y + b)
            J = (W @ J) * der[:, None]
        J = self.W_out @ J
        n = self.W_out.shape[0]
        return J[:, :n].copy(), J[:, n:].copy()


def synthetic_fnn(n=4, m=2, H=16, L=2, seed=0x5EED0004, act="relu"):
    """BASELINE config 4 model: weights ~ U(-1,1)/sqrt(fan_in) from the SplitMix64 generator
    (one stream per tensor); W_out is then rescaled so that the Jacobian at the origin has spectral radius 0.95
    (a slowly decaying plant; unscaled random weights give a radius of 0.04, i.e. nothing to control)."""
    def uni(stream, shape, fan_in):
        with np.errstate(over="ignore"):
            cnt = int(np.prod(shape))
            state = _mix64(np.uint64(seed & _MASK) + np.uint64(0x632BE59BD9B4E019) * np.uint64(stream + 1))
            out = np.empty(cnt)
            gamma = np.uint64(0x9E3779B97F4A7C15)
            for i in range(cnt):
                state = state + gamma
                out[i] = float(_mix64(state) >> np.uint64(11)) * 2.0 ** -53
        return (2.0 * out.reshape(shape) - 1.0) / math.sqrt(fan_in)
    W_in = uni(0, (H, n + m), n + m)
    W_h = [uni(1 + 2 * j, (H, H), H) for j in range(L)]
    b_h = [0.1 * uni(2 + 2 * j, (H,), H) * math.sqrt(H) for j in range(L)]
    W_out = uni(100, (n, H), H)
    model = FnnModel(W_in, W_h, b_h, W_out, act)
    A0, _ = model.jacobian(np.zeros(n), np.zeros(m))
    model.W_out = W_out * (0.95 / max(1e-12, float(np.max(np.abs(np.linalg.eigvals(A0))))))
    return model


def fnn_linear_problem(model: FnnModel, N, u_min, u_max, x_ref, u_ref, **kw):
    """The reference's LinearProgramming branch for a black-box model: dynamics linearised at the FIRST reference
    (.../fnn/...:38-46), terminal weight from the linearisation at the LAST reference (src/sub/design_mpc.jl:312-327),
    then exactly the linear path."""
    x_ref = np.asarray(x_ref, dtype=np.float64)
    u_ref = np.asarray(u_ref, dtype=np.float64)
    xr = x_ref.reshape(len(x_ref), -1)
    ur = u_ref.reshape(len(u_ref), -1)
    A, B = model.jacobian(xr[:, 0], ur[:, 0])
    Al, Bl = model.jacobian(xr[:, -1], ur[:, -1])
    n, m = B.shape
    q, r = kw.get("q", 100.0), kw.get("r", 0.1)
    P = dare(Al, Bl, q * np.eye(n), r * np.eye(m))
    return make_problem(A, B, N, u_min, u_max, x_ref=x_ref, u_ref=u_ref, P=P, **kw)


# --------------------------------------------------------------------------------------
# NLP branch for black-box models: the problem the reference hands to Ipopt
# (.../fnn/mpc_modeler_implementation_fnn.jl:73-189 + src/sub/design_mpc.jl:405-468)
# --------------------------------------------------------------------------------------
def fnn_rollout(model: FnnModel, x0, U):
    """x (n, N+1) of the network driven by U (m, N) from x0."""
    N = U.shape[1]
    X = np.empty((x0.size, N + 1))
    X[:, 0] = x0
    for k in range(N):
        X[:, k + 1] = model.forward(X[:, k], U[:, k])
    return X


def nlp_cost_and_gradient(model: FnnModel, x0, U, x_ref, u_ref, Q, R, S, P):
    """Single-shooting statement of the reference's NLP (states eliminated through the network): the cost of
    src/sub/design_mpc.jl:405-468 -- sum_{i=1..N} e_x_i'Q e_x_i + e_u_i'R e_u_i, terminal e_x_{N+1}'P e_x_{N+1}, rate term
    sum_{i<N} (u_i - u_{i+1})'S(u_i - u_{i+1}) on u itself -- and its exact gradient in U by the adjoint recursion.
    The R/S branch rules (only element [1,1] is tested, :423-465) are the caller's business: pass zeros to switch a term off."""
    n, N = x0.size, U.shape[1]
    X = fnn_rollout(model, x0, U)
    EX, EU = X - x_ref, U - u_ref
    J = float(EX[:, N] @ P @ EX[:, N])
    for k in range(N):
        J += float(EX[:, k] @ Q @ EX[:, k] + EU[:, k] @ R @ EU[:, k])
    for k in range(N - 1):
        du = U[:, k] - U[:, k + 1]
        J += float(du @ S @ du)
    G = np.zeros_like(U)
    lam = 2.0 * (0.5 * (P + P.T)) @ EX[:, N]
    Qs, Rs, Ss = 0.5 * (Q + Q.T), 0.5 * (R + R.T), 0.5 * (S + S.T)
    for k in range(N - 1, -1, -1):
        A, B = model.jacobian(X[:, k], U[:, k])
        G[:, k] = 2.0 * Rs @ EU[:, k] + B.T @ lam
        lam = A.T @ lam + (2.0 * Qs @ EX[:, k] if k > 0 else 0.0)
    for k in range(N - 1):
        du = 2.0 * Ss @ (U[:, k] - U[:, k + 1])
        G[:, k] += du
        G[:, k + 1] -= du
    return J, G, X


def nlp_kkt_residual(model: FnnModel, x0, U, x_ref, u_ref, Q, R, S, P, u_min, u_max):
    """Method-independent first-order certificate of a candidate U (m, N) for the box-constrained NLP: the projected-gradient
    residual |U - clip(U - D^2 G)|_inf in Jacobi-scaled coordinates (D^2 = 1 / diag of the Gauss-Newton Hessian would need the
    QP; the plain gradient scaled by 1/(2 R_aa) is used), zero exactly at KKT points."""
    _, G, _ = nlp_cost_and_gradient(model, x0, U, x_ref, u_ref, Q, R, S, P)
    sc = 1.0 / np.maximum(2.0 * np.diag(R), 1e-12)
    T = np.clip(U - sc[:, None] * G, u_min[:, None], u_max[:, None])
    return float(np.abs(U - T).max())


def sqp_fnn(model: FnnModel, x0, x_ref, u_ref, Q, R, S, P, u_min, u_max, iters, u_guess=None, step_scale=1.0, adaptive=False,
            merit_mu=None, structured=False, x_min=None, x_max=None, terminal="none", return_active=False):
    """CPU restatement of the device loop almpc_sqp_fnn_*: Gauss-Newton SQP with multiple shooting, every QP solved exactly
    (solve_box_qp_exact).  Returns X (n, N+1), U (m, N) and the per-iteration (|v|_inf, |defect|_inf) history (None for an
    iteration whose QP was void).
    adaptive: step rule 1 of almpc_sqp_fnn_set_step_rule -- l1 merit function phi = J + mu |defects|_1, mu = 2 max(|P|, |Q|),
    tested a posteriori at the top of the next iteration: if phi did not decrease the iterate returns to the last accepted point
    plus half the step and the iteration's QP is void; accepted steps double the factor back up to 1; at 1/64 accept anyway.
    x_min / x_max / terminal: state rows of every iteration's QP (ltv_state_rows); ValueError when a QP is infeasible."""
    m, N = u_ref.shape
    U = np.clip(u_ref if u_guess is None else u_guess, u_min[:, None], u_max[:, None]).astype(np.float64)
    X = fnn_rollout(model, x0, U)
    hist = []
    useR = R[0, 0] != 0.0
    Rz = R if useR else 0.0 * R
    Sz = S if (useR and S[0, 0] != 0.0) else 0.0 * S
    mu = 2.0 * max(np.abs(P).max(), np.abs(Q).max()) if merit_mu is None else float(merit_mu)
    a, ref = 1.0, np.inf
    Xb = Ub = dXb = Vb = None
    n_state_active = 0

    def merit(X, U, fv):
        EX, EU = X - x_ref, U - u_ref
        J = float(EX[:, N] @ P @ EX[:, N]) + sum(float(EX[:, k] @ Q @ EX[:, k] + EU[:, k] @ Rz @ EU[:, k]) for k in range(N))
        J += sum(float((U[:, k] - U[:, k + 1]) @ Sz @ (U[:, k] - U[:, k + 1])) for k in range(N - 1))
        return J + mu * float(np.abs(fv - X[:, 1:]).sum())

    for _ in range(iters):
        fv = np.stack([model.forward(X[:, k], U[:, k]) for k in range(N)], axis=1)
        if adaptive:
            phi = merit(X, U, fv)
            ok = (phi <= ref + 1e-12 * abs(ref) + 1e-300) or a <= 1.0 / 64.0
            if ok:
                ref, a = phi, min(1.0, 2.0 * a)
            else:
                a *= 0.5
                X = Xb + a * dXb
                U = np.clip(Ub + a * Vb, u_min[:, None], u_max[:, None])
                hist.append(None)
                continue
        A, B, c = [], [], []
        for k in range(N):
            Ak, Bk = model.jacobian(X[:, k], U[:, k])
            A.append(Ak); B.append(Bk); c.append(fv[:, k] - X[:, k + 1])
        if structured == "dual":   # the QP in its stage-wise form by the dual active set (k_sdual): also state rows and S
            import stagewise_oracle as so
            if x_min is not None and (np.any(x0 < x_min) or np.any(x0 > x_max)):
                raise ValueError("infeasible: x[:,1] = x0 violates the state box")
            qq = so.stage_qp_from_ltv(A, B, c, X, U, x_ref, u_ref, Q, Rz, Sz, P, u_min, u_max, x_min=x_min, x_max=x_max, terminal=terminal)
            rr = so.solve_stage_dual(qq, v_guess=np.zeros((N, m)))
            if rr["status"] == 3:
                raise ValueError("infeasible QP")
            v = rr["v"].T
            n_state_active = int(np.count_nonzero(rr["lam_x"]))
        elif structured:   # the QP in its stage-wise form (riccati_active_set): no condensed Hessian, also for unstable linearisations
            rr = riccati_active_set(A, B, Q, Rz, P, np.zeros(x0.size), (u_min[:, None] - U).T, (u_max[:, None] - U).T, c=np.array(c),
                                    ebar=(X[:, 1:] - x_ref[:, 1:]).T, qu=(Rz @ (U - u_ref)).T)
            v = rr["v"].T
        elif x_min is not None or terminal == "equality":
            if x_min is not None and (np.any(x0 < x_min) or np.any(x0 > x_max)):
                raise ValueError("infeasible: x[:,1] = x0 violates the state box")
            H, q, lo, hi, Gam, g = ltv_qp(A, B, c, X, U, x_ref, u_ref, Q, Rz, Sz, P, u_min, u_max, return_prediction=True)
            C, a0, lo_c, hi_c, eq_c = ltv_state_rows(Gam, g, X, x_ref, x_min, x_max, terminal)
            vv, W = solve_qp_rows_exact(H, q, lo, hi, C, a0, lo_c, hi_c, eq_c)
            n_state_active = sum(1 for j in W if j >= H.shape[0])
            v = vv.reshape(N, m).T
        else:
            H, q, lo, hi = ltv_qp(A, B, c, X, U, x_ref, u_ref, Q, Rz, Sz, P, u_min, u_max)
            v = solve_box_qp_exact(H, q, lo, hi).reshape(N, m).T
        hist.append((float(np.abs(v).max()), float(np.abs(np.array(c)).max())))
        dX = np.zeros_like(X)
        dx = np.zeros(x0.size)
        for k in range(N):
            dx = A[k] @ dx + B[k] @ v[:, k] + c[k]
            dX[:, k + 1] = dx
        if adaptive:
            Xb, Ub, dXb, Vb = X.copy(), U.copy(), step_scale * dX, step_scale * v
        sc = step_scale * (a if adaptive else 1.0)
        X, U = X + sc * dX, np.clip(U + sc * v, u_min[:, None], u_max[:, None])
    if return_active:
        return X, U, hist, n_state_active
    return X, U, hist


# --------------------------------------------------------------------------------------
# fixtures and synthetic configurations
# --------------------------------------------------------------------------------------
def decode_linear_regressor_fixture(blob: bytes):
    """(A, B) from the reference's test fixture test/models_saved/linear_regressor_train_result.jls
    (a Julia `Serialization` MLJ machine, 555 bytes).  The fitted 6x4 Float32 coefficient
    matrix sits at byte offset 221, column-major; the test builds AB = coefficients',
    A = AB[:,1:4], B = AB[:,5:6] (test/computation_mpc_test.jl:1003-1006)."""
    coef = np.frombuffer(blob, dtype="<f4", count=24, offset=221).reshape((6, 4), order="F")
    AB = coef.T.astype(np.float64)
    return AB[:, :4].copy(), AB[:, 4:].copy()


def qtp_linear_fixture_problem(A, B):
    """Scenario of test/computation_mpc_test.jl:981-1054: N=5, x_ref=0.65, u_ref=1.2,
    u in [0,4]x[0,3.26], defaults Q=100I, R=0.1I, terminal "none"; x0 = 0.6 (:1040).
    (`mpc_state_constraint` is not passed, so there is no state box.)"""
    return make_problem(A, B, 5, [0.0, 0.0], [4.0, 3.26], x_ref=[0.65] * 4, u_ref=[1.2] * 2)


def double_integrator(N=10):
    """BASELINE config 1: Ts=1, u in [-1,1], refs 0."""
    return make_problem([[1.0, 1.0], [0.0, 1.0]], [[0.5], [1.0]], N, [-1.0], [1.0])


def _expm(M):
    import scipy.linalg as sla
    return sla.expm(M)


def quadrotor_model(Ts=0.1, mass=0.5, J=(4e-3, 4e-3, 8e-3), g=9.81):
    """Hover-linearised quadrotor, states [p(3), v(3), (phi,theta,psi), omega(3)], inputs
    [dT, tau_x, tau_y, tau_z]; exact zero-order hold via expm([[Ac,Bc],[0,0]] Ts)
    (SURVEY.md section 8d config 2)."""
    Ac = np.zeros((12, 12))
    Bc = np.zeros((12, 4))
    Ac[0:3, 3:6] = np.eye(3)
    Ac[3, 7] = g
    Ac[4, 6] = -g
    Ac[6:9, 9:12] = np.eye(3)
    Bc[5, 0] = 1.0 / mass
    Bc[9, 1] = 1.0 / J[0]
    Bc[10, 2] = 1.0 / J[1]
    Bc[11, 3] = 1.0 / J[2]
    M = np.zeros((16, 16))
    M[:12, :12] = Ac
    M[:12, 12:] = Bc
    E = _expm(M * Ts)
    return E[:12, :12].copy(), E[:12, 12:].copy()


def quadrotor(N=30):
    """BASELINE config 2/3 controller: Q=100 I, R=0.1 I, S=0, P=DARE, input box
    [-2,3]x[+-0.05]x[+-0.05]x[+-0.02], references 0 (deviation coordinates)."""
    A, B = quadrotor_model()
    return make_problem(A, B, N, [-2.0, -0.05, -0.05, -0.02], [3.0, 0.05, 0.05, 0.02])


_MASK = (1 << 64) - 1


def _mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def splitmix_normal(seed: int, first_instance: int, count: int, dim: int):
    """Standard normals, shape (count, dim): one SplitMix64 stream per instance
    (state0 = mix64(seed + 0x632BE59BD9B4E019*(instance+1))), uniforms = top 53 bits,
    Box-Muller pairs (cos branch first).  Depends only on (seed, instance index), so a
    shard can generate its own slice of the batch."""
    with np.errstate(over="ignore"):
        inst = np.arange(first_instance, first_instance + count, dtype=np.uint64)
        state = _mix64(np.uint64(seed & _MASK) + np.uint64(0x632BE59BD9B4E019) * (inst + np.uint64(1)))
        npair = (dim + 1) // 2
        out = np.empty((count, 2 * npair))
        gamma = np.uint64(0x9E3779B97F4A7C15)
        for j in range(npair):
            state = state + gamma
            u1 = (_mix64(state) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
            state = state + gamma
            u2 = (_mix64(state) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
            r = np.sqrt(-2.0 * np.log(1.0 - u1))
            out[:, 2 * j] = r * np.cos(2.0 * np.pi * u2)
            out[:, 2 * j + 1] = r * np.sin(2.0 * np.pi * u2)
    return out[:, :dim]


QUADROTOR_X0_SCALE = np.array([1, 1, 1, .5, .5, .5, .1, .1, .1, .1, .1, .1])


def quadrotor_x0_batch(batch, amplitude=1.0, seed=0x5EED0002, first_instance=0):
    """x0_i = s * diag(1,1,1,.5,.5,.5,.1 x6) * xi_i, xi_i ~ N(0, I12).
    Returns (batch, 12), C-contiguous: instance-major, i.e. the C-ABI's [batch][state]."""
    xi = splitmix_normal(seed, first_instance, batch, 12)
    return np.ascontiguousarray(amplitude * xi * QUADROTOR_X0_SCALE[None, :])
