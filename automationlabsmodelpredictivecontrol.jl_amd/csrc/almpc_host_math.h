// almpc_host_math.h -- small dense column-major float64 helpers for the design-time host code
// (DARE on n x n, packing).  Not a CPU path of the solve: nothing here runs per step.
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

namespace almpc {
namespace hm {

typedef std::vector<double> mat;  // column-major, explicit dims at call sites

inline mat eye(int n) {
    mat I((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) I[(size_t)i * n + i] = 1.0;
    return I;
}

// C(r x c) = A(r x k) * B(k x c)
inline mat mul(const mat& A, const mat& B, int r, int k, int c) {
    mat C((size_t)r * c, 0.0);
    for (int j = 0; j < c; ++j)
        for (int l = 0; l < k; ++l) {
            const double b = B[(size_t)j * k + l];
            if (b == 0.0) continue;
            for (int i = 0; i < r; ++i) C[(size_t)j * r + i] += A[(size_t)l * r + i] * b;
        }
    return C;
}

inline mat transpose(const mat& A, int r, int c) {
    mat T((size_t)r * c);
    for (int j = 0; j < c; ++j)
        for (int i = 0; i < r; ++i) T[(size_t)i * c + j] = A[(size_t)j * r + i];
    return T;
}

// Solve A X = B (A n x n general, B n x c) by LU with partial pivoting. Returns false if singular.
inline bool lu_solve(mat A, mat& B, int n, int c) {
    std::vector<int> piv(n);
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::fabs(A[(size_t)k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(A[(size_t)k * n + i]) > best) { best = std::fabs(A[(size_t)k * n + i]); p = i; }
        if (!(best > 0.0)) return false;
        if (p != k) {
            for (int j = 0; j < n; ++j) std::swap(A[(size_t)j * n + k], A[(size_t)j * n + p]);
            for (int j = 0; j < c; ++j) std::swap(B[(size_t)j * n + k], B[(size_t)j * n + p]);
        }
        const double inv = 1.0 / A[(size_t)k * n + k];
        for (int i = k + 1; i < n; ++i) {
            const double f = A[(size_t)k * n + i] * inv;
            if (f == 0.0) continue;
            A[(size_t)k * n + i] = f;
            for (int j = k + 1; j < n; ++j) A[(size_t)j * n + i] -= f * A[(size_t)j * n + k];
            for (int j = 0; j < c; ++j) B[(size_t)j * n + i] -= f * B[(size_t)j * n + k];
        }
    }
    for (int j = 0; j < c; ++j)
        for (int i = n - 1; i >= 0; --i) {
            double s = B[(size_t)j * n + i];
            for (int l = i + 1; l < n; ++l) s -= A[(size_t)l * n + i] * B[(size_t)j * n + l];
            B[(size_t)j * n + i] = s / A[(size_t)i * n + i];
        }
    return true;
}

inline double amax(const mat& A) {  // max-abs norm (no squares: a diverging iterate must not overflow the test that rejects it)
    double s = 0.0;
    for (double v : A) s = std::fmax(s, std::fabs(v));
    return s;
}

// Stabilising solution of the discrete algebraic Riccati equation
//   A'PA - P - A'PB (R + B'PB)^-1 B'PA + Q = 0
// by the structure-preserving doubling algorithm.  Stands in for ControlSystems.are(Discrete, A, B, Q, R)
// (reference: src/sub/design_mpc.jl:327).
inline bool dare(const mat& A, const mat& B, const mat& Q, const mat& R, int n, int m, mat& P) {
    mat Rinv_Bt = transpose(B, n, m);  // m x n
    if (!lu_solve(R, Rinv_Bt, m, n)) return false;
    mat Ak = A, Gk = mul(B, Rinv_Bt, n, m, n), Hk = Q;
    const mat I = eye(n);
    bool converged = false;
    for (int it = 0; it < 200 && !converged; ++it) {
        mat W = mul(Gk, Hk, n, n, n);
        for (size_t i = 0; i < W.size(); ++i) W[i] += I[i];
        mat WinvA = Ak, WinvG = Gk;
        if (!lu_solve(W, WinvA, n, n)) return false;
        if (!lu_solve(W, WinvG, n, n)) return false;
        const mat At = transpose(Ak, n, n);
        mat A1 = mul(Ak, WinvA, n, n, n);
        mat G1 = mul(mul(Ak, WinvG, n, n, n), At, n, n, n);
        mat H1 = mul(mul(At, Hk, n, n, n), WinvA, n, n, n);
        for (size_t i = 0; i < G1.size(); ++i) { G1[i] += Gk[i]; H1[i] += Hk[i]; }
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < j; ++i) {
                const double h = 0.5 * (H1[(size_t)j * n + i] + H1[(size_t)i * n + j]);
                H1[(size_t)j * n + i] = H1[(size_t)i * n + j] = h;
                const double g = 0.5 * (G1[(size_t)j * n + i] + G1[(size_t)i * n + j]);
                G1[(size_t)j * n + i] = G1[(size_t)i * n + j] = g;
            }
        mat diff = H1;
        for (size_t i = 0; i < diff.size(); ++i) diff[i] -= Hk[i];
        const bool done = std::isfinite(amax(H1)) && amax(diff) <= 1e-13 * std::fmax(1.0, amax(H1));
        Ak.swap(A1); Gk.swap(G1); Hk.swap(H1);
        converged = done;
    }
    if (!converged) return false;  // 200 doublings without a fixed point: no stabilising solution to working precision
    for (double v : Hk)
        if (!std::isfinite(v)) return false;
    // accept only a P that satisfies the equation itself: residual of A'PA - P - A'PB (R + B'PB)^-1 B'PA + Q
    {
        const mat At = transpose(A, n, n), Bt = transpose(B, n, m);
        const mat PA = mul(Hk, A, n, n, n), PB = mul(Hk, B, n, n, m);
        mat S = mul(Bt, PB, m, n, m);
        for (size_t i = 0; i < S.size(); ++i) S[i] += R[i];
        mat K = mul(Bt, PA, m, n, n);  // B'PA, m x n
        if (!lu_solve(S, K, m, n)) return false;
        mat res = mul(At, PA, n, n, n);
        const mat corr = mul(mul(At, PB, n, n, m), K, n, m, n);
        for (size_t i = 0; i < res.size(); ++i) res[i] += Q[i] - Hk[i] - corr[i];
        if (!(amax(res) <= 1e-7 * std::fmax(1.0, std::fmax(amax(Hk), amax(Q))))) return false;
    }
    P = Hk;
    return true;
}

// Stage records of the UNCONSTRAINED stage-wise problem for k_sdual (csrc/almpc_sdual.hip.h), shared model: backward Riccati
// recursion of the reference's cost (src/sub/design_mpc.jl:405-468: Q on stages 1..N-1, P on stage N, R on every input, and -- when
// `S` is given -- the input-rate term (u_k - u_{k+1})'S(u_k - u_{k+1}), k = 0..N-2, which makes the stage state [e_k; v_{k-1}],
// nt = n + m, and adds cross terms) over e_{k+1} = A e_k + B v_k.  Per stage k and row l of NT + MC rows, 2 NT + 2 MC doubles:
//   l < nt:      [ Acl'(l, :) | -K(:, l)' | Acl(l, :) | -Bt(l, :) ]        Acl = At - Bt K,  K = Lam^-1 (Bt'P+At + M'),
//   l = NT + a:  [ Lb(a, :)   | Li(a, :)  | -K(a, :)  | -e_a      ]        Lam = Rt + Bt'P+Bt, Li = Lam^-1, Lb = Li Bt'
// (oracle: oracle/stagewise_oracle.py::stage_gains).  R: the reference's branch rule already applied by the caller (zeros when R[1,1]
// == 0).  `invariant`: every stage has the same record to 1e-11 (DARE terminal weight, no S): the kernel then loads it once.
inline bool stage_records(const mat& A, const mat& B, const mat& Q, const mat& R, const mat* S, const mat& P, int n, int m, int N,
                          int NT, int MC, mat& rec, bool& invariant) {
    const int nt = S ? n + m : n;
    if (nt > NT || m > MC) return false;
    const int RL = 2 * NT + 2 * MC, LRP = NT + MC;
    const size_t stage = (size_t)LRP * RL;
    rec.assign(stage * N, 0.0);
    mat At((size_t)nt * nt, 0.0), Bt((size_t)nt * m, 0.0);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) At[(size_t)j * nt + i] = A[(size_t)j * n + i];
    for (int a = 0; a < m; ++a) {
        for (int i = 0; i < n; ++i) Bt[(size_t)a * nt + i] = B[(size_t)a * n + i];
        if (S) Bt[(size_t)a * nt + n + a] = 1.0;
    }
    const mat AtT = transpose(At, nt, nt), BtT = transpose(Bt, nt, m);
    mat Pn((size_t)nt * nt, 0.0);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) Pn[(size_t)j * nt + i] = 0.5 * (P[(size_t)j * n + i] + P[(size_t)i * n + j]);
    for (int k = N - 1; k >= 0; --k) {
        mat Qt((size_t)nt * nt, 0.0), Mx((size_t)nt * m, 0.0), Rt = R;
        for (int j = 0; j < m; ++j)
            for (int i = 0; i < m; ++i) Rt[(size_t)j * m + i] = 0.5 * (R[(size_t)j * m + i] + R[(size_t)i * m + j]);
        if (k >= 1) {
            for (int j = 0; j < n; ++j)
                for (int i = 0; i < n; ++i) Qt[(size_t)j * nt + i] = 0.5 * (Q[(size_t)j * n + i] + Q[(size_t)i * n + j]);
            if (S)
                for (int b = 0; b < m; ++b)
                    for (int a = 0; a < m; ++a) {
                        const double sv = 0.5 * ((*S)[(size_t)b * m + a] + (*S)[(size_t)a * m + b]);
                        Qt[(size_t)(n + b) * nt + n + a] = sv;
                        Mx[(size_t)b * nt + n + a] = -sv;
                        Rt[(size_t)b * m + a] += sv;
                    }
        }
        const mat PA = mul(Pn, At, nt, nt, nt), PB = mul(Pn, Bt, nt, nt, m);
        mat Lam = mul(BtT, PB, m, nt, m);
        for (size_t i = 0; i < Lam.size(); ++i) Lam[i] += Rt[i];
        mat Gm = mul(BtT, PA, m, nt, nt);   // m x nt
        for (int j = 0; j < nt; ++j)
            for (int a = 0; a < m; ++a) Gm[(size_t)j * m + a] += Mx[(size_t)a * nt + j];
        mat Li = eye(m);
        if (!lu_solve(Lam, Li, m, m)) return false;
        for (int b = 0; b < m; ++b)
            for (int a = 0; a < b; ++a) { const double v = 0.5 * (Li[(size_t)b * m + a] + Li[(size_t)a * m + b]); Li[(size_t)b * m + a] = Li[(size_t)a * m + b] = v; }
        const mat K = mul(Li, Gm, m, m, nt);     // m x nt
        const mat Lb = mul(Li, BtT, m, m, nt);   // m x nt
        mat Acl = At;
        const mat BK = mul(Bt, K, nt, m, nt);
        for (size_t i = 0; i < Acl.size(); ++i) Acl[i] -= BK[i];
        for (double v : K)
            if (!std::isfinite(v)) return false;
        double* rk = rec.data() + stage * k;
        for (int l = 0; l < nt; ++l) {
            double* row = rk + (size_t)l * RL;
            for (int j = 0; j < nt; ++j) row[j] = Acl[(size_t)l * nt + j];                       // Acl'(l, j) = Acl(j, l)
            for (int b = 0; b < m; ++b) row[NT + b] = -K[(size_t)l * m + b];                    // -K(b, l)
            for (int j = 0; j < nt; ++j) row[NT + MC + j] = Acl[(size_t)j * nt + l];            // Acl(l, j)
            for (int b = 0; b < m; ++b) row[2 * NT + MC + b] = -Bt[(size_t)b * nt + l];         // -Bt(l, b)
        }
        for (int a = 0; a < m; ++a) {
            double* row = rk + (size_t)(NT + a) * RL;
            for (int j = 0; j < nt; ++j) row[j] = Lb[(size_t)j * m + a];
            for (int b = 0; b < m; ++b) row[NT + b] = Li[(size_t)b * m + a];
            for (int j = 0; j < nt; ++j) row[NT + MC + j] = -K[(size_t)j * m + a];
            row[2 * NT + MC + a] = -1.0;
        }
        // P_k = Qt + At'P+At - Gm'K, symmetrised
        mat Pnew = mul(AtT, PA, nt, nt, nt);
        const mat GK = mul(transpose(Gm, m, nt), K, nt, m, nt);
        for (size_t i = 0; i < Pnew.size(); ++i) Pnew[i] += Qt[i] - GK[i];
        for (int j = 0; j < nt; ++j)
            for (int i = 0; i <= j; ++i) {
                const double v = 0.5 * (Pnew[(size_t)j * nt + i] + Pnew[(size_t)i * nt + j]);
                Pn[(size_t)j * nt + i] = Pn[(size_t)i * nt + j] = v;
            }
    }
    double scale = 0.0, dev = 0.0;
    for (size_t i = 0; i < stage; ++i) scale = std::fmax(scale, std::fabs(rec[stage * (N - 1) + i]));
    for (int k = 0; k + 1 < N; ++k)
        for (size_t i = 0; i < stage; ++i) dev = std::fmax(dev, std::fabs(rec[stage * k + i] - rec[stage * (N - 1) + i]));
    invariant = dev <= 1e-11 * std::fmax(scale, 1e-300);
    return true;
}

}  // namespace hm
}  // namespace almpc
