// almpc_design.hip.h -- design-time kernels: prediction matrices, condensed Hessian, shared inverses.
//
// Reference functions restated (paths relative to /root/reference):
//   dynamics e_x[:,k+1] = A e_x[:,k] + B e_u[:,k]   src/sub/model_modeler_implementation/linear/mpc_modeler_implementation_linear.jl:58-60
//   objective                                        src/sub/design_mpc.jl:405-468
//   terminal weight P on stage N+1                   src/sub/design_mpc.jl:327,448-456
// condensed as in SURVEY.md section 8a:  E = Phi e0 + Gamma v,  H = 2(Gamma' Qbar Gamma + Rbar + D'Sbar D),
// F = 2 Gamma' Qbar Phi.  Kernels (SURVEY.md Appendix B ids):
//   k_design_blocks   K1/K2  Phi_k = A^(k+1), G_k = A^k B
//   k_design_gamma           Gamma, W = Qbar Gamma, WP = Qbar Phi   (row-major, zero padded)
//   k_design_hessian  K3/K4  H and F tiles by FP64 MFMA over LDS-staged row panels (the B'QB contraction)
//   k_design_scale           d = diag(H)^-1/2, H' = DHD, F' = DF
//   k_design_inverse  K5     (H' + cI)^-1 by in-LDS Cholesky, triangular inverse, X'X
//   k_pack_frags             dense -> MFMA A-fragment layout used by k_admm
#pragma once
#include "almpc_kernels.hip.h"

#include <string>
#include <vector>

namespace almpc {

// Every design kernel takes the instance number from blockIdx.y and per-instance strides (in doubles) for its operands:
// all zero with gridDim.y = 1 for the shared-model design, the per-instance sizes for almpc_design_batched.
struct DesignStrides {
    long A = 0, B = 0, P = 0, Phi = 0, Gk = 0, Gam = 0, WP = 0, H = 0, F = 0, d = 0, Hs = 0, Fs = 0, G = 0, Minv = 0, rho = 0,
         flag = 0;
    int h_symmetric = 0;   // H is symmetric by construction (k_design_instance): k_design_scale skips its transposed, uncoalesced read
};

// ---- K1/K2 -------------------------------------------------------------------------------------
// One workgroup.  Phi[k] (n x n, column-major) = A^(k+1); Gk[k] (n x m, column-major) = A^k B.
inline __global__ __launch_bounds__(256) void k_design_blocks(int n, int m, int N, const double* A, const double* B,
                                                       double* Phi, double* Gk, DesignStrides st) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    A += blockIdx.y * st.A; B += blockIdx.y * st.B; Phi += blockIdx.y * st.Phi; Gk += blockIdx.y * st.Gk;
    double* As = smem;           // n*n
    double* Bs = As + n * n;     // n*m
    double* cur = Bs + n * m;    // n*n  = A^k
    double* nxt = cur + n * n;
    for (int t = threadIdx.x; t < n * n; t += blockDim.x) {
        As[t] = A[t];
        cur[t] = (t % n == t / n) ? 1.0 : 0.0;
    }
    for (int t = threadIdx.x; t < n * m; t += blockDim.x) Bs[t] = B[t];
    __syncthreads();
    for (int k = 0; k < N; ++k) {
        for (int t = threadIdx.x; t < n * m; t += blockDim.x) {
            const int i = t % n, j = t / n;
            double s = 0.0;
            for (int l = 0; l < n; ++l) s += cur[l * n + i] * Bs[j * n + l];
            Gk[(size_t)k * n * m + t] = s;
        }
        for (int t = threadIdx.x; t < n * n; t += blockDim.x) {
            const int i = t % n, j = t / n;
            double s = 0.0;
            for (int l = 0; l < n; ++l) s += As[l * n + i] * cur[j * n + l];
            nxt[t] = s;
            Phi[(size_t)k * n * n + t] = s;
        }
        __syncthreads();
        double* tmp = cur; cur = nxt; nxt = tmp;
    }
}

// ---- Gamma, W = Qbar Gamma, WP = Qbar Phi -------------------------------------------------------
// Block k = stage k (rows k*n .. k*n+n-1).  Row-major with strides gs (Gam, W) and ps (WP); buffers are
// zero-initialised by the host so padding rows/columns stay zero.
inline __global__ __launch_bounds__(256) void k_design_gamma(int n, int m, int N, const double* Q, const double* P,
                                                      const double* Phi, const double* Gk, double* Gam, double* W,
                                                      double* WP, int gs, int ps, DesignStrides st) {
    const int k = blockIdx.x;
    P += blockIdx.y * st.P; Phi += blockIdx.y * st.Phi; Gk += blockIdx.y * st.Gk;
    Gam += blockIdx.y * st.Gam; W += blockIdx.y * st.Gam; WP += blockIdx.y * st.WP;
    const double* Qk = (k == N - 1) ? P : Q;  // stage N+1 carries only P (src/sub/design_mpc.jl:448-456)
    const int nz = m * N;
    for (int t = threadIdx.x; t < n * nz; t += blockDim.x) {
        const int p = t / nz, c = t % nz;
        const int i = c / m, a = c % m;
        double g = 0.0, w = 0.0;
        if (k >= i) {
            const double* G = Gk + (size_t)(k - i) * n * m + (size_t)a * n;
            g = G[p];
            for (int q2 = 0; q2 < n; ++q2) w += Qk[(size_t)q2 * n + p] * G[q2];
        }
        Gam[(size_t)(k * n + p) * gs + c] = g;
        W[(size_t)(k * n + p) * gs + c] = w;
    }
    for (int t = threadIdx.x; t < n * n; t += blockDim.x) {
        const int p = t / n, j = t % n;
        double w = 0.0;
        for (int q2 = 0; q2 < n; ++q2) w += Qk[(size_t)q2 * n + p] * Phi[(size_t)k * n * n + (size_t)j * n + q2];
        WP[(size_t)(k * n + p) * ps + j] = w;
    }
}

// ---- K3/K4: H = 2 Gamma' W (+ 2 Rbar + 2 D'Sbar D), F = 2 Gamma' WP ------------------------------
// Workgroup I = row block of 16 condensed variables.  Wave J < nrb computes the H tile (I, J); wave
// J >= nrb computes F tile (I, J - nrb).  The contraction index (rows of Gamma, n*N of them) is walked in
// panels of KC rows staged through LDS; panels above the block-triangular zero region are skipped.
struct HessParams {
    int n, m, N, nz, nrb, njf;  // njf = ceil(n/16)
    int kr;                     // padded contraction length (multiple of KC)
    const double* Gam; const double* W; const double* WP;
    int gs, ps;
    const double* R; const double* S;
    int useR, useS;
    double* H;  // column-major nz x nz
    double* F;  // column-major nz x n
    DesignStrides st;
};

constexpr int HESS_KC = 32;

inline __global__ __launch_bounds__(768) void k_design_hessian(HessParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    p.Gam += blockIdx.y * p.st.Gam; p.W += blockIdx.y * p.st.Gam; p.WP += blockIdx.y * p.st.WP;
    p.H += blockIdx.y * p.st.H; p.F += blockIdx.y * p.st.F;
    const int ws = p.gs + 16, pps = p.ps + 16;  // padded LDS strides: the 4 k-groups of a B read hit different banks
    double* GI = smem;                       // [KC][16]
    double* Wc = GI + HESS_KC * 16;          // [KC][ws]
    double* WPc = Wc + HESS_KC * ws;         // [KC][pps]
    const int I = blockIdx.x;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 15, q = lane >> 4;
    const int J = wv;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    // Gamma[:, 16I..16I+15] is zero for rows of stages < floor(16 I / m)
    const int first_row = (16 * I / p.m) * p.n;
    const int kstart = (first_row / HESS_KC) * HESS_KC;
    for (int kc = kstart; kc < p.kr; kc += HESS_KC) {
        for (int t = threadIdx.x; t < HESS_KC * 16; t += blockDim.x) {
            const int r = t / 16, c = t % 16;
            GI[t] = p.Gam[(size_t)(kc + r) * p.gs + I * 16 + c];
        }
        for (int t = threadIdx.x; t < HESS_KC * p.gs; t += blockDim.x) {
            const int r = t / p.gs, c = t % p.gs;
            Wc[r * ws + c] = p.W[(size_t)(kc + r) * p.gs + c];
        }
        for (int t = threadIdx.x; t < HESS_KC * p.ps; t += blockDim.x) {
            const int r = t / p.ps, c = t % p.ps;
            WPc[r * pps + c] = p.WP[(size_t)(kc + r) * p.ps + c];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < HESS_KC / 4; ++s) {
            const double a = GI[(4 * s + q) * 16 + col];
            const double b = (J < p.nrb) ? Wc[(4 * s + q) * ws + J * 16 + col] : WPc[(4 * s + q) * pps + (J - p.nrb) * 16 + col];
            acc = mfma_f64(a, b, acc);
        }
        __syncthreads();
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int r = I * 16 + q + 4 * reg;
        if (r >= p.nz) continue;
        if (J < p.nrb) {
            const int c = J * 16 + col;
            if (c >= p.nz) continue;
            double v = 2.0 * acc[reg];
            const int ir = r / p.m, ar = r % p.m, ic = c / p.m, ac = c % p.m;
            if (p.useR && ir == ic) v += 2.0 * p.R[(size_t)ac * p.m + ar];
            if (p.useS) {  // delta_u[:,i] = u[:,i] - u[:,i+1], i = 1..N-1 (src/sub/design_mpc.jl:429-431)
                if (ir == ic) {
                    const int cnt = (ir <= p.N - 2 ? 1 : 0) + (ir >= 1 ? 1 : 0);
                    v += 2.0 * cnt * p.S[(size_t)ac * p.m + ar];
                } else if (ir - ic == 1 || ic - ir == 1) {
                    v -= 2.0 * p.S[(size_t)ac * p.m + ar];
                }
            }
            p.H[(size_t)c * p.nz + r] = v;
        } else {
            const int j = (J - p.nrb) * 16 + col;
            if (j < p.n) p.F[(size_t)j * p.nz + r] = 2.0 * acc[reg];
        }
    }
}

// ---- scaling -------------------------------------------------------------------------------------
// d = diag(H)^-1/2 (pad rows 1), Hs (ld = nzs, symmetrised) = D H D, Fs (ld = nzs) = D F.  flag[0] != 0 on a
// non-positive diagonal.
// (a device function: the producers of H_i whose workgroup holds one instance -- k_design_instance_t -- run it as their own tail on the
// H_i they have just written, one launch less per design; ds: 128 doubles of LDS)
__device__ __forceinline__ void design_scale_body(int nz, int nzs, int n, const double* H, const double* F, double* d, double* Hs,
                                                  double* Fs, int* flag, int h_symmetric, double* ds) {
    for (int t = threadIdx.x; t < nzs; t += blockDim.x) {
        double v = 1.0;
        if (t < nz) {
            const double h = H[(size_t)t * nz + t];
            if (!(h > 0.0)) { atomicExch(flag, 1); v = 1.0; }
            else v = 1.0 / sqrt(h);
        }
        d[t] = v;
        if (t < 128) ds[t] = v;
    }
    __syncthreads();
    if (h_symmetric && (nz & 1) == 0) {
        // H_i came out symmetric (LDS route of k_design_instance) and columns are 16-byte aligned: two rows per load and store, two pairs
        // per thread and pass, (column, row pair) carried along instead of divided out
        const int hp = nz >> 1, total = hp * nz, step = 2 * (int)blockDim.x, dc = step / hp, dr = step - dc * hp;
        int t0 = 2 * threadIdx.x, c0 = t0 / hp, r0 = t0 - c0 * hp;
        for (; t0 < total; t0 += step) {
            d2 hv[2];
            int rr[2], cc[2];
            int r = r0, c = c0;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const bool in = t0 + u < total;
                cc[u] = in ? c : nz - 1; rr[u] = in ? r : hp - 1;
                hv[u] = *reinterpret_cast<const d2*>(H + (size_t)cc[u] * nz + 2 * rr[u]);
                if (++r == hp) { r = 0; ++c; }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (t0 + u < total) {
                    d2 o;
                    o[0] = ds[2 * rr[u]] * hv[u][0] * ds[cc[u]];
                    o[1] = ds[2 * rr[u] + 1] * hv[u][1] * ds[cc[u]];
                    *reinterpret_cast<d2*>(Hs + (size_t)cc[u] * nzs + 2 * rr[u]) = o;
                }
            r0 += dr; c0 += dc;
            if (r0 >= hp) { r0 -= hp; ++c0; }
        }
    } else
    // four consecutive elements per thread and pass, (column, row) carried along instead of divided out (one division per thread)
    {
        const int step = 4 * (int)blockDim.x, dc = step / nz, dr = step - dc * nz;
        int t0 = 4 * threadIdx.x, c0 = t0 / nz, r0 = t0 - c0 * nz;
        for (; t0 < nz * nz; t0 += step) {
            double ha[4], hb[4];
            int rr[4], cc[4];
            int r = r0, c = c0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in = t0 + u < nz * nz;
                cc[u] = in ? c : nz - 1; rr[u] = in ? r : nz - 1;
                ha[u] = H[(size_t)cc[u] * nz + rr[u]];
                hb[u] = h_symmetric ? ha[u] : H[(size_t)rr[u] * nz + cc[u]];   // (0.5 (a + a) = a exactly)
                if (++r == nz) { r = 0; ++c; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (t0 + u < nz * nz) Hs[(size_t)cc[u] * nzs + rr[u]] = ds[rr[u]] * (0.5 * (ha[u] + hb[u])) * ds[cc[u]];
            r0 += dr; c0 += dc;
            if (r0 >= nz) { r0 -= nz; ++c0; }
        }
    }
    for (int t = threadIdx.x; t < nz * n; t += blockDim.x) {
        const int r = t % nz, c = t / nz;
        Fs[(size_t)c * nzs + r] = ds[r] * F[(size_t)c * nz + r];
    }
}

inline __global__ __launch_bounds__(256) void k_design_scale(int nz, int nzs, int n, const double* H, const double* F,
                                                      double* d, double* Hs, double* Fs, int* flag, DesignStrides st) {
    H += blockIdx.y * st.H; F += blockIdx.y * st.F; d += blockIdx.y * st.d; Hs += blockIdx.y * st.Hs; Fs += blockIdx.y * st.Fs;
    flag += blockIdx.y * st.flag;
    __shared__ double ds[128];  // nz <= 128 (almpc_create); the scaling is read nz^2 times
    design_scale_body(nz, nzs, n, H, F, d, Hs, Fs, flag, st.h_symmetric, ds);
}

// ---- K5: Out = (Hs + c I + diag(dshift))^-1 ----------------------------------------------------------------
// One workgroup (256 threads) per matrix, the matrix in REGISTERS as 4 x 16 tiles: thread (ri = tid & 31, g = tid >> 5) holds
// S[ri + 32 a][16 g + k], a < 4, k < 16.  In-place Gauss-Jordan sweeps without pivoting (SPD: every pivot is a Schur-complement
// diagonal, positive; flag[0] = 2 otherwise).  Per pivot the owners publish the pivot column and row through LDS (double
// buffered: one barrier per pivot); a thread then reads 16 row and 4 column entries for its 64 FMAs -- the kernel is bound by
// the LDS return path, and the 4 x 16 tile moves 3.2 x fewer bytes per FMA than one row x 32 columns per thread did (same
// operations on the same operands: bit-identical results).  nz <= 128.
constexpr int DESIGN_INVERSE_THREADS = 256;
// <AR, KC>: row slots and columns per thread -- the register tile covers 32 AR x 8 KC entries: <4, 16> = 128 x 128 (nz <= 128), <2, 8> =
// 64 x 64 (nz <= 64: a quarter of the FMAs per pivot; the re-linearisation pipeline's 40 x 40 matrices).  Same operations on the same
// operands in both: bit-identical results.
// Round 2, measured on 4096 quadrotor-size matrices (1.24 ms per launch, SQ counters: 106 VALU instructions per pivot and thread for
// 32 FMAs, VALU 57 % busy, 26 % of the wave cycles in s_waitcnt) and NOT faster, each tried by itself: 512 threads with a 4 x 8 tile
// (4 waves per SIMD instead of 2), the reciprocal by v_rcp_f64 + two Newton steps instead of the IEEE division (kept: it is shorter),
// a wave-uniform column ownership with the pivot row eliminated like the others, pivots unrolled in windows of four with the
// register tile rotated in between (loop body 10 KB of code instead of 39 KB: instruction fetch is not the limit either).  Also
// built, measured and removed: a blocked sweep on
// FP64 MFMA (8 x 8 tiles of 16 x 16 as accumulators, four waves, pivot block inverted inside one wave by cross-lane reads, T = A_IK Pi
// with one step of iterative refinement -- without it the explicit 16 x 16 inverse costs five digits: 4e-7 against 1.7e-11 --, row K
// from the transposes of the column results, two barriers per block step): 1.4 ms per launch, slower than this kernel, and 8e-8 on u
// against 1e-9.  The per-tile-row chain (five layout changes through LDS and twelve dependent MFMAs before the update can start)
// is as latency bound as the pivots here.
template <int AR, int KC, int NG = 8>   // NG column groups of 32 threads each
__global__ __launch_bounds__(32 * NG) void k_design_inverse_t(int nz, int nzs, const double* Hs, double cshift, const double* dshift,
                                                        double* Out, int* flag, long sHs, long sShift, long sOut, long sFlag) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    Hs += blockIdx.y * sHs; Out += blockIdx.y * sOut; flag += blockIdx.y * sFlag;
    if (dshift) dshift += blockIdx.y * sShift;
    double* prow = smem;            // [2][128] pivot row (raw)
    double* pcol = smem + 256;      // [2][128] pivot column
    int* badp = reinterpret_cast<int*>(smem + 512);
    const int ri = threadIdx.x & 31, g = threadIdx.x >> 5;
    if (threadIdx.x == 0) *badp = 0;
    double S[AR][KC];
    // all loads in flight before the first wait: clamped addresses instead of a branch per element (see k_design_inverse_wave)
    double shiftv[AR];
#pragma unroll
    for (int a = 0; a < AR; ++a) {
        const int i = ri + 32 * a, ic = i < nz ? i : nz - 1;
        shiftv[a] = cshift + (dshift ? dshift[ic] : 0.0);
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int c = KC * g + k;
            S[a][k] = Hs[(size_t)(c < nz ? c : nz - 1) * nzs + ic];
        }
    }
#pragma unroll
    for (int a = 0; a < AR; ++a)
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int i = ri + 32 * a, c = KC * g + k;
            S[a][k] = (i < nz && c < nz) ? S[a][k] + (i == c ? shiftv[a] : 0.0) : 0.0;
        }
    // The reciprocal of a pivot is a chain of five dependent FP64 operations (v_rcp_f64 + two Newton steps: ~550 cycles of latency,
    // measured with the stamps below) that every thread used to walk AFTER the barrier, i.e. on the critical path of every pivot.
    // It is computed one pivot early instead: while pivot pv updates the tile, every thread updates the column of pivot pv + 1 first
    // and starts the reciprocal of its own entry of that column in the row slot of row pv + 1 -- for the thread that owns element
    // (pv + 1, pv + 1) that is the next pivot -- under the other 60 FMAs of the update; the owner publishes it with the pivot row.
    auto fast_rcp = [](double v) __attribute__((always_inline)) -> double {
        const double pvs = v > 0.0 ? v : 1.0;
        double r = __builtin_amdgcn_rcp(pvs);
        r = __builtin_fma(__builtin_fma(-pvs, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-pvs, r, 1.0), r, r);
        return r;
    };
    double* ipslot = smem + 514;   // [2] reciprocal of the pivot, published by its owner
    double nip = fast_rcp(S[0][0]);   // (pivot 0: thread 0's entry)
    double ncand = S[0][0];
#ifdef ALMPC_STAMPS
    long long sg_[6] = {0, 0, 0, 0, 0, 0}, st_ = 0;
#define INV_T(S_) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = __builtin_readcyclecounter(); if (S_ >= 0) sg_[S_ < 0 ? 0 : S_] += t_ - st_; st_ = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define INV_T(S_)
#endif
    for (int gq = 0; gq < NG; ++gq) {
        const int ap = (KC * gq) >> 5;              // row slot of the pivot rows of this column group (uniform)
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int pv = KC * gq + k;
            if (pv < nz) {  // uniform
                double* pr = prow + (pv & 1) * 128;
                double* pc = pcol + (pv & 1) * 128;
                const bool rowp = ri == (pv & 31);  // this thread holds a piece of row pv (in slot ap)
                INV_T(-1);
                if (g == gq) {                      // column pv, every row
#pragma unroll
                    for (int a = 0; a < AR; ++a) pc[ri + 32 * a] = S[a][k];
                }
                if (rowp) {                         // row pv, this thread's KC columns
#pragma unroll
                    for (int a = 0; a < AR; ++a)
                        if (a == ap) {
#pragma unroll
                            for (int kk = 0; kk < KC; kk += 2) {
                                d2 w; w[0] = S[a][kk]; w[1] = S[a][kk + 1];
                                *reinterpret_cast<d2*>(pr + KC * g + kk) = w;
                            }
                        }
                    if (g == gq) {                  // this thread owns the pivot element itself
                        ipslot[pv & 1] = nip;
                        if (!(ncand > 0.0)) *badp = 1;
                    }
                }
                INV_T(0);
                __syncthreads();
                INV_T(1);
#ifdef ALMPC_STAMPS
                if (g_stamps && threadIdx.x == 0 && (blockIdx.y == 0 || blockIdx.y == gridDim.y / 2))
                    g_stamps[(size_t)(blockIdx.y == 0 ? 0 : 1) * 256 + pv] = __builtin_readcyclecounter();
#endif
                const double ip = ipslot[pv & 1];
                INV_T(2);
                double w[KC];
#pragma unroll
                for (int kk = 0; kk < KC; kk += 2) {
                    const d2 t = *reinterpret_cast<const d2*>(pr + KC * g + kk);
                    w[kk] = t[0]; w[kk + 1] = t[1];
                }
                INV_T(3);
                double f[AR];
                bool ispa[AR];
#pragma unroll
                for (int a = 0; a < AR; ++a) {
                    ispa[a] = rowp && a == ap;
                    f[a] = ispa[a] ? 0.0 : pc[ri + 32 * a] * ip;   // the pivot row itself is rescaled, not eliminated
                }
                // the column of the next pivot first, and its reciprocal in flight under the rest of the update
                const int kn = (k + 1) % KC;   // (a constant once the pivot loop is unrolled)
                const int an = (pv + 1) >> 5;
#pragma unroll
                for (int a = 0; a < AR; ++a) {
                    double v = __builtin_fma(-f[a], w[kn], S[a][kn]);
                    if (ispa[a]) v *= ip;
                    S[a][kn] = v;
                    if (a == an) ncand = v;
                }
                nip = fast_rcp(ncand);
#pragma unroll
                for (int a = 0; a < AR; ++a) {
#pragma unroll
                    for (int kk = 0; kk < KC; ++kk)
                        if (kk != kn) S[a][kk] = __builtin_fma(-f[a], w[kk], S[a][kk]);
                    if (ispa[a]) {
#pragma unroll
                        for (int kk = 0; kk < KC; ++kk)
                            if (kk != kn) S[a][kk] *= ip;
                    }
                    if (g == gq) S[a][k] = ispa[a] ? ip : -f[a];  // column pv
                }
                INV_T(4);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < AR; ++a)
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int i = ri + 32 * a, c = KC * g + k;
            if (i < nz && c < nz) Out[(size_t)c * nzs + i] = S[a][k];
        }
    __syncthreads();
    if (threadIdx.x == 0 && *badp) atomicExch(flag, 2);
#ifdef ALMPC_STAMPS
    if (g_stamps && threadIdx.x == 0 && blockIdx.y == 0)
        for (int c_ = 0; c_ < 5; ++c_) g_stamps[(size_t)2 * 256 + c_] = sg_[c_];
#endif
}
// ---- K5, nz <= 64: ONE WAVE per matrix, no barrier ------------------------------------------------------------------------------
// Lane i holds row i (NCOL >= nz columns in registers).  A pivot is S_ij -= f_i w_j with f = pivot column / d (each lane's own
// register) and w = pivot row, which every lane needs as NCOL scalars.  Round 2 fetched them with v_readlane (an SGPR operand per
// FMA): 2 x v_readlane_b32 + FMA per column measured 38-44 cycles per column whatever the order (tools/microbench/inverse_wave.hip:
// a VALU write to a scalar register costs far more than its issue slot), 1.7-2.4 k cycles per pivot.  Now the pivot row comes
// through LDS with no cross-lane instruction at all:
//  * the sweep keeps the matrix (anti)symmetric -- S_pj = S_jp for columns not yet used as pivots, -S_jp for the ones that were --
//    so the pivot ROW is the pivot COLUMN with those signs: one ds_write_b64 per lane publishes it (lane j writes w_j), NCOL / 2
//    ds_read_b128 at a wave-uniform address bring it back as VGPR operands, one v_fma_f64 per column and nothing else;
//  * the pivot row is rescaled by the SAME FMA that eliminates the other rows: its lane takes f = 1 - 1/d, so that
//    fma(-f, w_j, S_pj) = S_pj / d (the multiplier 1 - f carries one extra rounding of size eps d: nothing for d <= 1, which is every
//    pivot of a Jacobi-scaled Hessian without shift; 1.7 digits on that one row for the largest shifted pivot, d = 46, of the
//    ADMM's KKT matrix, which only feeds the working-set guess);
//  * pivots are unrolled in blocks of eight with compile-time register indices and the register row is rotated by eight columns per
//    block (position c holds column (c + pivots done, rounded down to eight) mod NCOL); the pivot row is published in position
//    order, so the reads have immediate offsets;
//  * the column of the next pivot is updated first, its pivot row is published and its reciprocal started under the other FMAs.
// The wave talks to itself only: LDS operations of one wave execute in order, the fence keeps the compiler from reordering them.
#ifdef INV_STAMPS
__device__ long long g_inv_dbg[16];
#define INV_W(N_) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_inv_dbg[N_] = __builtin_readcyclecounter(); } while (0)
#else
#define INV_W(N_)
#endif
// rho_G != null: the penalty profile of the ADMM is made HERE instead of by a launch of its own (k_design_rho: 4.6 us of a 190 us
// re-linearisation step): rho_i = rho_mode ? rho / G_ii : rho from the inverse rho_G of the first call, written to rho_out and used as
// the diagonal shift of this one.
template <int NCOL>
__global__ __launch_bounds__(256) void k_design_inverse_wave(int nz, int nzs, int batch, const double* Hs, double cshift, const double* dshift,
                                                             double* Out, int* flag, long sHs, long sShift, long sOut, long sFlag,
                                                             const double* rho_G = nullptr, long sRhoG = 0, int rho_mode = 0, double rho = 0.0,
                                                             double* rho_out = nullptr, const double* vM = nullptr, double* vOut = nullptr,
                                                             long sV = 0, int vcols = 0, double* Out2 = nullptr, long sOut2 = 0,
                                                             double cshift2 = 0.0, int vld = 0) {
    constexpr int PB = 8;
    static_assert(NCOL % PB == 0 && NCOL > PB, "column count: a multiple of the pivot block");
    __shared__ __attribute__((aligned(16))) double wbuf_all[4][2][64];
    const int lane = threadIdx.x & 63;
    int inst = blockIdx.x * 4 + (threadIdx.x >> 6);
    // Out2 != null: BOTH inverses of a per-instance design in one launch (scalar rho: the second does not need the first) -- waves
    // batch .. 2 batch - 1 form Out2 = (Hs + (cshift2 + rho) I)^-1 and write the (constant) penalty profile; a wave per matrix is
    // latency bound, 1024 of them are one per SIMD: the second inverse rides in the empty issue slots instead of taking its own 28 us
    const bool second = Out2 != nullptr && inst >= batch;
    if (second) inst -= batch;
    if (inst >= batch) return;
    double* wbuf = &wbuf_all[threadIdx.x >> 6][0][0];
    Hs += (size_t)inst * sHs; flag += (size_t)inst * sFlag;
    if (second) { Out = Out2 + (size_t)inst * sOut2; cshift = cshift2; vM = nullptr; }
    else Out += (size_t)inst * sOut;
    if (dshift) dshift += (size_t)inst * sShift;
    const int i = lane;
    double S[NCOL];
    // every load is issued before the first one is waited for: the addresses are clamped into the matrix instead of branched around
    // (a branch per element made the compiler wait for each load in turn: NCOL memory latencies in a row, 20 of the 27 us this
    // kernel took for 40 x 40 matrices)
    INV_W(0);
    const int ic = i < nz ? i : nz - 1;
    double shift = cshift + (dshift ? dshift[ic] : 0.0);
    if (second) {
        if (i < nzs) rho_out[(size_t)inst * sShift + i] = i < nz ? rho : 1.0;   // (pad rows 1, as k_design_rho)
        shift = cshift + rho;
    } else if (rho_G) {
        const double gii = rho_G[(size_t)inst * sRhoG + (size_t)ic * nzs + ic];
        const double rv = rho_mode == 1 ? rho / gii : rho;
        if (i < nzs) rho_out[(size_t)inst * sShift + i] = i < nz ? rv : 1.0;   // (pad rows 1, as k_design_rho)
        shift = cshift + rv;
    }
#pragma unroll
    for (int c = 0; c < NCOL; ++c) S[c] = Hs[(size_t)(c < nz ? c : nz - 1) * nzs + ic];
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        const double v = S[c] + (i == c ? shift : 0.0);
        S[c] = (i < nz && c < nz) ? v : ((i == c) ? 1.0 : 0.0);   // (padding: unit diagonal)
    }
    auto fast_rcp = [](double v) __attribute__((always_inline)) -> double {
        const double pvs = v > 0.0 ? v : 1.0;
        double r = __builtin_amdgcn_rcp(pvs);
        r = __builtin_fma(__builtin_fma(-pvs, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-pvs, r, 1.0), r, r);
        return r;
    };
    // lane j publishes w_j of pivot pv (its entry of column pv, sign by the rule above) at the POSITION of column j
    auto publish = [&](double colv, int pv, int rot) __attribute__((always_inline)) {
        int pos = i - rot; if (pos < 0) pos += NCOL;
        if (i < NCOL) wbuf[(pv & 1) * 64 + pos] = (i < pv) ? -colv : colv;
    };
    bool bad = false;
    double piv = readlane_d(S[0], 0);
    INV_W(1);
    double ip = fast_rcp(piv);
    int nrot = 0;
    publish(S[0], 0, 0);
#pragma unroll 1
    for (int p0 = 0; p0 < nz; p0 += PB) {
        INV_W(4 + (p0 >> 3));
#pragma unroll
        for (int k = 0; k < PB; ++k) {
            const int pv = p0 + k;
            if (pv < nz) {   // (uniform)
                if (!(piv > 0.0)) bad = true;
#ifdef INV_STAMPS
#define INV_P(N_) do { if (k == 3 && p0 == 8) { __builtin_amdgcn_sched_barrier(0); INV_W(N_); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define INV_P(N_)
#endif
                INV_P(10);
                wave_fence_lds();
                const double* w = wbuf + (pv & 1) * 64;
                const bool isp = i == pv;
                const double f = isp ? 1.0 - ip : S[k] * ip;
                // the next pivot first: column at position k + 1, row pv + 1
                const int pn = pv + 1 < 64 ? pv + 1 : 63;
                S[k + 1] = __builtin_fma(-f, w[k + 1], S[k + 1]);
                publish(S[k + 1], pv + 1, k + 1 == PB ? nrot + PB : nrot);
#ifdef INV_EXP_NOCHAIN
                const double npiv = piv + 1e-9;
                const double nip = ip + 1e-9;
#else
                INV_P(11);
                const double npiv = readlane_d(S[k + 1], pn);
                INV_P(12);
                const double nip = fast_rcp(npiv);
                INV_P(13);
#endif
#ifdef INV_EXP_NOFMA
                if (pv < 0)
#endif
#pragma unroll
                for (int c = 0; c < NCOL; c += 2) {
                    const d2 wc = *reinterpret_cast<const d2*>(w + c);
                    if (c != k && c != k + 1) S[c] = __builtin_fma(-f, wc[0], S[c]);
                    if (c + 1 != k && c + 1 != k + 1) S[c + 1] = __builtin_fma(-f, wc[1], S[c + 1]);
                }
                S[k] = isp ? ip : -f;            // column pv of the inverse so far
                piv = npiv; ip = nip;
                INV_P(14);
            }
        }
        // rotate by one block: the next eight pivot columns move to positions 0..7, the finished ones to the end
        double t[PB];
#pragma unroll
        for (int k = 0; k < PB; ++k) t[k] = S[k];
#pragma unroll
        for (int c = 0; c + PB < NCOL; ++c) S[c] = S[c + PB];
#pragma unroll
        for (int k = 0; k < PB; ++k) S[NCOL - PB + k] = t[k];
        nrot += PB;
    }
    INV_W(2);
    // position c holds column (c + nrot) mod NCOL
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
        const int col = (c + nrot) % NCOL;
        if (i < nz && col < nz) Out[(size_t)col * nzs + i] = S[c];
    }
    if (__any(bad) && lane == 0) atomicExch(flag, 2);
    // vM != null: V = -Out vM (nz x vcols, leading dimension nzs: the per-instance V_i = -G_i F'_i of a design) from the rows this wave
    // still holds, instead of a launch of its own that reads G_i back (k_neg_gm_cols: 7 us of a 186 us re-linearisation step).  Column
    // by column: lane j publishes vM[j, c] at the position of column j, every lane reads the NCOL values back at wave-uniform
    // addresses -- the pivot-row mechanism once more.
    if (vM) {
        const double* Mi = vM + (size_t)inst * sV;
        double* Vi = vOut + (size_t)inst * sV;
        int pos = i - (nrot % NCOL); if (pos < 0) pos += NCOL;
        for (int c = 0; c < vcols; ++c) {
            const double mv = Mi[(size_t)c * vld + ic];
            wave_fence_lds();
            if (i < NCOL) wbuf[pos] = i < nz ? mv : 0.0;
            wave_fence_lds();
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int q = 0; q < NCOL; q += 2) {
                const d2 w2 = *reinterpret_cast<const d2*>(wbuf + q);
                a0 = __builtin_fma(S[q], w2[0], a0);
                a1 = __builtin_fma(S[q + 1], w2[1], a1);
            }
            if (i < nz) Vi[(size_t)c * vld + i] = -(a0 + a1);
        }
    }
    INV_W(3);
}

// ---- K5, 64 < nz <= 128: four waves per matrix, wave q = columns 32 q .. 32 q + 31 of ALL rows (lane l: rows l and l + 64) --------
// The one-wave scheme above with the columns dealt to four waves.  Per pivot ONE barrier and one publication: the wave that owns the
// pivot column writes w (the pivot row, = its column with the signs of the sweep's antisymmetry) and the reciprocal of the pivot,
// which the lane holding it has computed itself -- no v_readlane anywhere; every wave then reads its two f sources (its own rows of
// w), 16 ds_read_b128 at a wave-uniform address for its 32 columns, and issues 64 FMAs.  The column of the next pivot is updated and
// published first, under the other columns.  Against the 4 x 16 register tile (k_design_inverse_t: two barrier round trips and a pivot
// column AND row per pivot, 2.3-2.8 k cycles per pivot): 112 -> 41 us for 256 matrices of 100 x 100 (the SQP iteration), 1.31 -> 0.6 ms
// for 4096 of 120 x 120 (the batched design).
// HOIST: all sixteen pivot-row reads of a wave are issued right behind the two reads of the chain (one matrix per CU: their latency
// runs under the chain, 65 -> 58 us for 256 matrices of 100 x 100); without it they are issued four at a time between the FMAs (two
// or more workgroups per CU: a chain read does not queue behind 8 waves x 16 reads in the CU's one LDS pipeline, 0.63 against 1.03 ms
// for 4096 matrices of 120 x 120).
// CW: columns per wave (128 / CW waves per matrix).
template <int CW, bool HOIST>
__global__ __launch_bounds__(64 * (128 / CW)) void k_design_inverse_c32(int nz, int nzs, const double* Hs, double cshift, const double* dshift,
                                                            double* Out, int* flag, long sHs, long sShift, long sOut, long sFlag,
                                                            const double* vM = nullptr, double* vOut = nullptr, long sV = 0, int vcols = 0,
                                                            int vld = 0, int packed_out = 0) {
    // packed_out: Out is the packed lower triangle of the (symmetric) inverse, column by column (packed_tri_off; sOut = its stride):
    // what k_admm_inst<true> streams -- half the bytes written here and read there
    __shared__ __attribute__((aligned(16))) double wb[2][128];
    __shared__ double vred[128 / CW][128];   // (vM != null: the waves' partial sums of V = -Out vM)
    __shared__ double ips[2];
    __shared__ int badp;
    Hs += blockIdx.y * sHs; Out += blockIdx.y * sOut; flag += blockIdx.y * sFlag;
    if (dshift) dshift += blockIdx.y * sShift;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    if (threadIdx.x == 0) badp = 0;
    double S[2][CW];
    double shiftv[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {   // every load in flight before the first wait (clamped addresses, no branch per element)
        const int i = lane + 64 * a, ic = i < nz ? i : nz - 1;
        shiftv[a] = cshift + (dshift ? dshift[ic] : 0.0);
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const int col = CW * q + c;
            S[a][c] = Hs[(size_t)(col < nz ? col : nz - 1) * nzs + ic];
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const int i = lane + 64 * a, col = CW * q + c;
            S[a][c] = (i < nz && col < nz) ? S[a][c] + (i == col ? shiftv[a] : 0.0) : (i == col ? 1.0 : 0.0);
        }
    auto fast_rcp = [](double v) __attribute__((always_inline)) -> double {
        const double pvs = v > 0.0 ? v : 1.0;
        double r = __builtin_amdgcn_rcp(pvs);
        r = __builtin_fma(__builtin_fma(-pvs, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-pvs, r, 1.0), r, r);
        return r;
    };
    if (q == 0) {   // pivot 0: column 0 as it is
        wb[0][lane] = S[0][0]; wb[0][lane + 64] = S[1][0];
        if (lane == 0) { ips[0] = fast_rcp(S[0][0]); if (!(S[0][0] > 0.0)) badp = 1; }
    }
#pragma unroll 1
    for (int qp = 0; qp < 128 / CW; ++qp) {
#pragma unroll
        for (int k = 0; k < CW; ++k) {
            const int pv = CW * qp + k;
            if (pv < nz) {   // (uniform)
#ifdef INV_STAMPS
#define INV_Q(N_) do { if (k == 5 && qp == 0) { __builtin_amdgcn_sched_barrier(0); INV_W(N_); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define INV_Q(N_)
#endif
                INV_Q(9);
                __syncthreads();
                INV_Q(10);
                // every LDS read of this pivot is issued here, ahead of the publication below (the compiler will not move a read
                // across a write to the other half of the buffer): their latency runs under the chain of the next pivot
                const double* w = wb[pv & 1];
                const double ip = ips[pv & 1];
                double wi[2];
                // (relaxed atomic loads = plain ds_read_b64 that the optimiser does not sink into a branch behind the sixteen others)
                wi[0] = __hip_atomic_load(wb[pv & 1] + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                wi[1] = __hip_atomic_load(wb[pv & 1] + lane + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                double wreg[CW];
                if (HOIST) {
#pragma unroll
                    for (int c = 0; c < CW; c += 2) {
                        const d2 t = *reinterpret_cast<const d2*>(w + CW * q + c);
                        wreg[c] = t[0]; wreg[c + 1] = t[1];
                    }
                }
                double f[2];
                bool isp[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int i = lane + 64 * a;
                    isp[a] = i == pv;
                    f[a] = isp[a] ? 1.0 - ip : ((i < pv) ? -wi[a] : wi[a]) * ip;   // S_i,pv / d; the pivot row is rescaled by the same FMA
                }
                INV_Q(11);
                // the column of the next pivot first (local index kn in the wave that owns it; an ordinary column in the others)
                const int kn = (k + 1) % CW;
                const int owner_next = (k + 1 < CW) ? qp : qp + 1;
                if (!HOIST) wreg[kn] = w[CW * q + kn];
#pragma unroll
                for (int a = 0; a < 2; ++a) S[a][kn] = __builtin_fma(-f[a], wreg[kn], S[a][kn]);
#ifdef INV_EXP_NOPUBLISH
                if (q == owner_next && pv + 1 < nz && pv < 0) {
#else
                if (q == owner_next && pv + 1 < nz) {
#endif
                    double* wn = wb[(pv + 1) & 1];
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const int i = lane + 64 * a;
                        wn[i] = (i <= pv) ? -S[a][kn] : S[a][kn];
                    }
                    const double cand = ((pv + 1) >> 6) ? S[1][kn] : S[0][kn];
                    const double r = fast_rcp(cand);
                    if (lane == ((pv + 1) & 63)) { ips[(pv + 1) & 1] = r; if (!(cand > 0.0)) badp = 1; }
                }
                INV_Q(12);
                // the other columns
#ifdef INV_EXP_NOBULK
                if (pv < 0)
#endif
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    if (!HOIST && (c & 1) == 0) {
                        const d2 t = *reinterpret_cast<const d2*>(w + CW * q + c);
                        wreg[c] = t[0]; wreg[c + 1] = t[1];
                    }
                    if (c == kn) continue;
                    if (c == k && q == qp) {   // column pv of the inverse so far
#pragma unroll
                        for (int a = 0; a < 2; ++a) S[a][k] = isp[a] ? ip : -f[a];
                        continue;
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a) S[a][c] = __builtin_fma(-f[a], wreg[c], S[a][c]);
                }
                INV_Q(13);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const int i = lane + 64 * a, col = CW * q + c;
            if (packed_out) {
                if (i < nz && col <= i) Out[packed_tri_off(nz, col) + (i - col)] = S[a][c];
                // (nz even: an even column is followed by one pad entry; it keeps a copy of M[col+1][col], which lets the reader
                // of the diagonal block take its off-diagonal entry from either side -- see k_admm_inst<true>)
                if (i < nz && i == col + 1 && ((nz - col) & 1) == 0) Out[packed_tri_off(nz, col) + (nz - col)] = S[a][c];
            }
            else if (i < nz && col < nz) Out[(size_t)col * nzs + i] = S[a][c];
        }
    // vM != null: V = -Out vM (nz x vcols, leading dimension vld) from the columns the waves still hold -- wave q sums its CW columns,
    // the partial sums meet in LDS -- instead of a launch of its own that reads the inverse back (k_neg_gm / k_neg_gm_cols)
    if (vM) {
        const double* Mi = vM + blockIdx.y * sV;
        double* Vi = vOut + blockIdx.y * sV;
        for (int vc = 0; vc < vcols; ++vc) {
            double p0 = 0.0, p1 = 0.0;
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const int col = CW * q + c;
                const double mv = col < nz ? Mi[(size_t)vc * vld + col] : 0.0;   // (wave-uniform address)
                p0 = __builtin_fma(S[0][c], mv, p0);
                p1 = __builtin_fma(S[1][c], mv, p1);
            }
            vred[q][lane] = p0; vred[q][lane + 64] = p1;
            __syncthreads();
            if (q == 0) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int i = lane + 64 * a;
                    double sum = 0.0;
#pragma unroll
                    for (int w = 0; w < 128 / CW; ++w) sum += vred[w][i];
                    if (i < nz) Vi[(size_t)vc * vld + i] = -sum;
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && badp) atomicExch(flag, 2);
}

// launcher: the smallest register tile that holds the matrix
// whether launch_design_inverse makes the ADMM penalty profile itself (rho_G given): the one-wave kernel does, the others need k_design_rho first
inline bool design_inverse_makes_rho(int nz, int nzs) { return nz <= 64 && nzs <= 64 && !getenv("ALMPC_INV_TILE") && !getenv("ALMPC_NO_RHO_FUSION"); }
// (the one-wave-per-matrix kernel can also form V = -Out vM from the rows it holds: see there)
// whether launch_design_inverse can write the packed lower triangle (the column-split kernel, 64 < nz <= 128)
inline bool design_inverse_can_pack(int nz) { return nz > 64 && nz <= 128 && (nz & 1) == 0 && !getenv("ALMPC_INV_TILE") && !getenv("ALMPC_NO_PACKED_MINV"); }
inline bool design_inverse_makes_v(int nz) { return nz <= 128 && !getenv("ALMPC_INV_TILE") && !getenv("ALMPC_DBG_SPLIT_NEGGM"); }   // (one-wave and column-split kernels)
inline void launch_design_inverse(dim3 grid, size_t lds, hipStream_t st, int nz, int nzs, const double* Hs, double cshift, const double* dshift,
                                  double* Out, int* flag, long sHs, long sShift, long sOut, long sFlag,
                                  const double* rho_G = nullptr, long sRhoG = 0, int rho_mode = 0, double rho = 0.0, double* rho_out = nullptr,
                                  const double* vM = nullptr, double* vOut = nullptr, long sV = 0, int vcols = 0,
                                  double* Out2 = nullptr, long sOut2 = 0, double cshift2 = 0.0, int vld = 0, int packed_out = 0) {
    // (packed_out: only the column-split kernel writes the packed triangle; the caller asks design_inverse_can_pack first)
    if (vld == 0) vld = nzs;
    if (nz <= 64 && !getenv("ALMPC_INV_TILE")) {   // one wave per matrix (grid.y = matrices, as for the tile kernels)
        const int b = (int)grid.y;
        const dim3 g2((unsigned)(((Out2 ? 2 * b : b) + 3) / 4));
#define INV_WAVE(NC_) hipLaunchKernelGGL((k_design_inverse_wave<NC_>), g2, dim3(256), 0, st, nz, nzs, b, Hs, cshift, dshift, Out, flag, sHs, sShift, sOut, sFlag, rho_G, sRhoG, rho_mode, rho, rho_out, vM, vOut, sV, vcols, Out2, sOut2, cshift2, vld)
        if (nz <= 16) INV_WAVE(16); else if (nz <= 32) INV_WAVE(32); else if (nz <= 48) INV_WAVE(48); else INV_WAVE(64);
#undef INV_WAVE
    } else if (nz <= 64) hipLaunchKernelGGL((k_design_inverse_t<2, 8>), grid, dim3(DESIGN_INVERSE_THREADS), lds, st, nz, nzs, Hs, cshift, dshift, Out, flag, sHs, sShift, sOut, sFlag);
    else if (!getenv("ALMPC_INV_TILE")) {
        const char* cwv = getenv("ALMPC_INV_CW");   // (experiment: columns per wave)
        const int cw = cwv ? atoi(cwv) : (grid.y <= 256 ? 16 : 32);   // one matrix per CU: eight waves of 16 columns (51 against 59 us at nz 100)
#define INV_C32(CW_, H_) hipLaunchKernelGGL((k_design_inverse_c32<CW_, H_>), grid, dim3(64 * (128 / CW_)), 0, st, nz, nzs, Hs, cshift, dshift, Out, flag, sHs, sShift, sOut, sFlag, vM, vOut, sV, vcols, vld, packed_out)
        if (grid.y <= 256) { if (cw == 8) INV_C32(8, true); else if (cw == 16) INV_C32(16, true); else INV_C32(32, true); }
        else { if (cw == 8) INV_C32(8, false); else if (cw == 16) INV_C32(16, false); else INV_C32(32, false); }
#undef INV_C32
    }
    else hipLaunchKernelGGL((k_design_inverse_t<4, 16>), grid, dim3(DESIGN_INVERSE_THREADS), lds, st, nz, nzs, Hs, cshift, dshift, Out, flag, sHs, sShift, sOut, sFlag);
}

// ---- K5 (shared-model design): Out = (Hs + c I + diag(dshift))^-1 by Cholesky -----------------------------
// The shared design runs once, so it takes the more accurate route (relative error ~ eps cond instead of the
// Gauss-Jordan kernel's few-times-larger constant: 4.5e-12 against 2.9e-10 in bench.py's u_err_inf).
// One workgroup, matrix in LDS (ld = nz+1 to spread banks).  Cholesky (right-looking), in-place inverse of
// the triangular factor, then X'X.  flag[0] != 0 if a pivot is not positive.
inline __global__ __launch_bounds__(512) void k_design_inverse_chol(int nz, int nzs, const double* Hs, double cshift, const double* dshift,
                                                        double* Out, int* flag, long sHs, long sShift, long sOut, long sFlag) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    Hs += blockIdx.y * sHs; Out += blockIdx.y * sOut; flag += blockIdx.y * sFlag;
    if (dshift) dshift += blockIdx.y * sShift;
    const int ld = nz + 1;
    double* L = smem;         // [nz][ld], element (i,j) at L[j*ld + i]
    double* xo = L + (size_t)nz * ld;  // [nz] scratch column
    int* badp = reinterpret_cast<int*>(xo + nz);  // in the dynamic region: a static __shared__ would misalign it
    if (threadIdx.x == 0) *badp = 0;
    for (int t = threadIdx.x; t < nz * nz; t += blockDim.x) {
        const int i = t % nz, j = t / nz;
        L[j * ld + i] = Hs[(size_t)j * nzs + i] + (i == j ? cshift + (dshift ? dshift[i] : 0.0) : 0.0);
    }
    __syncthreads();
    for (int j = 0; j < nz; ++j) {
        const double piv = L[j * ld + j];
        if (!(piv > 0.0)) { if (threadIdx.x == 0) *badp = 1; }
        const double ljj = sqrt(piv > 0.0 ? piv : 1.0);
        __syncthreads();
        for (int i = j + threadIdx.x; i < nz; i += blockDim.x) L[j * ld + i] = (i == j) ? ljj : L[j * ld + i] / ljj;
        __syncthreads();
        const int rem = nz - j - 1;
        for (int t = threadIdx.x; t < rem * rem; t += blockDim.x) {
            const int i = j + 1 + t % rem, k = j + 1 + t / rem;
            if (i >= k) L[k * ld + i] -= L[j * ld + i] * L[j * ld + k];
        }
        __syncthreads();
    }
    // in-place inverse of the lower-triangular factor (unblocked dtrti2, lower): columns right to left
    for (int j = nz - 1; j >= 0; --j) {
        const double ajj = 1.0 / L[j * ld + j];
        // x := T x with T = already inverted trailing block, x = L[j+1:, j]
        for (int i = j + 1 + threadIdx.x; i < nz; i += blockDim.x) {
            double s = 0.0;
            for (int k = j + 1; k <= i; ++k) s += L[k * ld + i] * L[j * ld + k];
            xo[i] = -s * ajj;
        }
        __syncthreads();
        for (int i = j + threadIdx.x; i < nz; i += blockDim.x) L[j * ld + i] = (i == j) ? ajj : xo[i];
        __syncthreads();
    }
    // Out = X'X, X = L^-1 (lower): Out[i][j] = sum_{k >= max(i,j)} X[k][i] X[k][j]
    for (int t = threadIdx.x; t < nz * nz; t += blockDim.x) {
        const int i = t % nz, j = t / nz;
        if (i < j) continue;
        double s = 0.0;
        for (int k = i; k < nz; ++k) s += L[i * ld + k] * L[j * ld + k];
        Out[(size_t)j * nzs + i] = s;
        Out[(size_t)i * nzs + j] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0 && *badp) atomicExch(flag, 2);
}

// ---- ADMM penalty per row: scalar rho (OSQP), or the stiffness profile rho_i = rho / G_ii (G = H'^-1), pad rows 1
inline __global__ __launch_bounds__(256) void k_design_rho(int nz, int nzs, int mode, double rho, const double* G, double* rhovec,
                                                    long sG, long sRho) {
    G += blockIdx.y * sG; rhovec += blockIdx.y * sRho;
    for (int t = threadIdx.x; t < nzs; t += blockDim.x)
        rhovec[t] = (t < nz) ? (mode == 1 ? rho / G[(size_t)t * nzs + t] : rho) : 1.0;
}

// ---- dense (column-major, ld) -> MFMA A-fragment layout ---------------------------------------------
// frag[(rb*ks + s)*64 + l] = M[rb*16 + (l&15)][4 s + (l>>4)]  (zero outside rows x cols)
inline __global__ __launch_bounds__(256) void k_pack_frags(const double* M, int rows, int cols, int ld, int nrb, int ks,
                                                    double* frag) {
    const int total = nrb * ks * 64;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int l = t & 63, s = (t >> 6) % ks, rb = (t >> 6) / ks;
        const int r = rb * 16 + (l & 15), c = 4 * s + (l >> 4);
        frag[t] = (r < rows && c < cols) ? M[(size_t)c * ld + r] : 0.0;
    }
}

// ---- Out[:, c] = -G M[:, c] for a few columns (leading dimension ld for M and Out; G dense symmetric, leading dimension nzs).
// Design-time / set_reference-time helper for the polish's unconstrained minimiser v0 = -G f' = (-G F') e0 + (-G fS):
// with V = -G F' packed like F', k_admm gets v0 from n columns instead of a second nz x nz product per step.
inline __global__ __launch_bounds__(256) void k_neg_gm(int nz, int nzs, int ncols, int ld, const double* G, const double* M, double* Out,
                                                long sG, long sM) {
    G += blockIdx.y * sG; M += blockIdx.y * sM; Out += blockIdx.y * sM;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < (long)ncols * nz; t += (long)gridDim.x * blockDim.x) {
        const int c = (int)(t / nz), r = (int)(t % nz);
        const double* mcol = M + (size_t)c * ld;
        // G symmetric: consecutive r, consecutive addresses.  Four partial sums and eight loads in flight: the plain loop was one
        // dependent load-FMA per row of G (27 us for a 100 x 100 product per instance in the SQP loop)
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int j = 0;
        for (; j + 8 <= nz; j += 8) {
            double g[8], mv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { g[u] = G[(size_t)(j + u) * nzs + r]; mv[u] = mcol[j + u]; }
            a0 += g[0] * mv[0] + g[4] * mv[4];
            a1 += g[1] * mv[1] + g[5] * mv[5];
            a2 += g[2] * mv[2] + g[6] * mv[6];
            a3 += g[3] * mv[3] + g[7] * mv[7];
        }
        for (; j < nz; ++j) a0 += G[(size_t)j * nzs + r] * mcol[j];
        Out[(size_t)c * ld + r] = -((a0 + a1) + (a2 + a3));
    }
}

// The same product for ALL columns of one instance at a time (per-instance designs: V_i = -G_i F'_i, n columns): k_neg_gm reads G once
// per column -- n x 123 KB per quadrotor-size instance through L2 -- this one reads it once: thread = row r, the n columns'
// partial sums in registers, M staged in LDS.  Same summation order per element as k_neg_gm (bit-identical results).
template <int NC>
__global__ __launch_bounds__(128) void k_neg_gm_cols(int nz, int nzs, int ncols, int ld, const double* G, const double* M, double* Out,
                                                     long sG, long sM) {
    extern __shared__ __attribute__((aligned(16))) double smem[];   // [ncols][nz]
    G += blockIdx.y * sG; M += blockIdx.y * sM; Out += blockIdx.y * sM;
    for (int t = threadIdx.x; t < ncols * nz; t += blockDim.x) smem[t] = M[(size_t)(t / nz) * ld + t % nz];
    __syncthreads();
    const int r = threadIdx.x < nz ? threadIdx.x : nz - 1;
    double a[NC][4];
#pragma unroll
    for (int c = 0; c < NC; ++c) { a[c][0] = 0.0; a[c][1] = 0.0; a[c][2] = 0.0; a[c][3] = 0.0; }
    int j = 0;
    for (; j + 8 <= nz; j += 8) {
        double g[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) g[u] = G[(size_t)(j + u) * nzs + r];
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < ncols) {
                const double* mv = smem + c * nz + j;
                a[c][0] += g[0] * mv[0] + g[4] * mv[4];
                a[c][1] += g[1] * mv[1] + g[5] * mv[5];
                a[c][2] += g[2] * mv[2] + g[6] * mv[6];
                a[c][3] += g[3] * mv[3] + g[7] * mv[7];
            }
    }
    for (; j < nz; ++j) {
        const double gv = G[(size_t)j * nzs + r];
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < ncols) a[c][0] += gv * smem[c * nz + j];
    }
    if (threadIdx.x < nz) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < ncols) Out[(size_t)c * ld + r] = -((a[c][0] + a[c][1]) + (a[c][2] + a[c][3]));
    }
}
// per-instance V_i = -G_i F'_i: all n columns in one pass over G_i when they fit the register tile
inline void launch_neg_gm_batched(hipStream_t st, unsigned batch, int nz, int nzs, int ncols, const double* G, const double* M, double* Out,
                                  long sG, long sM) {
    const size_t lds = (size_t)ncols * nz * sizeof(double);
    if (nz <= 128 && ncols <= 4 && lds <= 48 * 1024) hipLaunchKernelGGL((k_neg_gm_cols<4>), dim3(1, batch), dim3(128), lds, st, nz, nzs, ncols, nzs, G, M, Out, sG, sM);
    else if (nz <= 128 && ncols <= 8 && lds <= 48 * 1024) hipLaunchKernelGGL((k_neg_gm_cols<8>), dim3(1, batch), dim3(128), lds, st, nz, nzs, ncols, nzs, G, M, Out, sG, sM);
    else if (nz <= 128 && ncols <= 16 && lds <= 48 * 1024) hipLaunchKernelGGL((k_neg_gm_cols<16>), dim3(1, batch), dim3(128), lds, st, nz, nzs, ncols, nzs, G, M, Out, sG, sM);
    else hipLaunchKernelGGL(k_neg_gm, dim3(4, batch), dim3(256), 0, st, nz, nzs, ncols, nzs, G, M, Out, sG, sM);
}

// ---- constraint space for state rows (state box / terminal equality): Ghat = A G A',  A = [I; C'],  C' = Gamma[rows] D ----
// Cold path: one thread per output element.
inline __global__ __launch_bounds__(256) void k_build_cprime(int mc, int nz, int nzs, int gs, const int* rowsel, const double* Gam,
                                                      const double* d, double* Cp) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < mc * nzs; t += gridDim.x * blockDim.x) {
        const int r = t / nzs, c = t % nzs;
        Cp[t] = (c < nz) ? Gam[(size_t)rowsel[r] * gs + c] * d[c] : 0.0;
    }
}
// CG[r][c] = sum_j Cp[r][j] G[j][c]
inline __global__ __launch_bounds__(256) void k_cg(int mc, int nz, int nzs, const double* Cp, const double* G, double* CG) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < mc * nzs; t += gridDim.x * blockDim.x) {
        const int r = t / nzs, c = t % nzs;
        double s = 0.0;
        if (c < nz)
            for (int j = 0; j < nz; ++j) s += Cp[(size_t)r * nzs + j] * G[(size_t)j * nzs + c];
        CG[t] = s;
    }
}
inline __global__ __launch_bounds__(256) void k_ghat(int nz, int mc, int nzs, int Rs, const double* G, const double* Cp,
                                              const double* CG, double* Ghat, double* gnorm) {
    const int R = nz + mc;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < R * Rs; t += gridDim.x * blockDim.x) {
        const int a = t / Rs, b = t % Rs;
        double v = 0.0;
        if (b < R) {
            if (a < nz && b < nz) v = G[(size_t)a * nzs + b];
            else if (a < nz) v = CG[(size_t)(b - nz) * nzs + a];
            else if (b < nz) v = CG[(size_t)(a - nz) * nzs + b];
            else
                for (int j = 0; j < nz; ++j) v += CG[(size_t)(a - nz) * nzs + j] * Cp[(size_t)(b - nz) * nzs + j];
        }
        Ghat[t] = v;
        if (a == b) gnorm[a] = sqrt(v > 0.0 ? v : 1.0);
    }
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Rs; t += gridDim.x * blockDim.x)
        if (t >= R) gnorm[t] = 1.0;
}

// Terminal equality eliminated at design time: Ghat -= GhatE' Y with Y = Ghat_EE^-1 GhatE (host), rows / columns E exactly zero,
// gnorm from the projected diagonal (1 on E).  One thread per element.
inline __global__ __launch_bounds__(256) void k_ghat_project(int R, int Rs, int ne, int eq0, const double* GhatE, const double* Y, double* Ghat,
                                                      double* gnorm) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < R * Rs; t += gridDim.x * blockDim.x) {
        const int a = t / Rs, b = t % Rs;
        if (b >= R) continue;
        double v = Ghat[t];
        for (int e = 0; e < ne; ++e) v -= GhatE[(size_t)e * Rs + a] * Y[(size_t)e * Rs + b];
        if ((a >= eq0 && a < eq0 + ne) || (b >= eq0 && b < eq0 + ne)) v = 0.0;
        Ghat[t] = v;
        if (a == b) gnorm[a] = (a >= eq0 && a < eq0 + ne) ? 1.0 : sqrt(v > 0.0 ? v : 1.0);
    }
}

// Runs the whole shared-model design on `stream` and leaves the packed operands in the handle's buffers.
// Returns 0 or a negative almpc_status value (numbers as in include/almpc.h).
inline int design_shared_device(hipStream_t stream, int n, int m, int N, int nzs, int nrb, int ks, int ksf,
                                const std::vector<double>& A, const std::vector<double>& B, std::vector<double> Q,
                                std::vector<double> R, std::vector<double> S, std::vector<double> P, double rho,
                                double sigma, double* dMinvFrag, double* dVFrag, double* dHFrag, double* dFFrag,
                                double* dG, double* dD, std::vector<double>& hH, std::vector<double>& hF,
                                std::vector<double>& hd, std::string& err, const std::vector<int>& rowsel = std::vector<int>(),
                                int Rs = 0, double* dGhat = nullptr, double* dGnorm = nullptr, int rho_mode = 0,
                                double* dRho = nullptr, double* dVsOut = nullptr, double* dPlainOut = nullptr) {
    // dPlainOut (or null): [Minv nz x nzs | H' nz x nzs | F' n x nzs | V n x nzs], the dense column-major operands before they are packed
    // into MFMA fragments -- what the one-wave-per-instance step of small shared problems reads (k_step_inst_wave with stride 0)
    const int nz = m * N;
    const int useR = R[0] != 0.0, useS = useR && S[0] != 0.0;  // the reference tests only element [1,1]
    auto symmetrise = [](std::vector<double>& M, int k) {
        for (int j = 0; j < k; ++j)
            for (int i = 0; i < j; ++i) {
                const double v = 0.5 * (M[(size_t)j * k + i] + M[(size_t)i * k + j]);
                M[(size_t)j * k + i] = M[(size_t)i * k + j] = v;
            }
    };
    symmetrise(Q, n); symmetrise(P, n); symmetrise(R, m); symmetrise(S, m);
    const int njf = (n + 15) / 16, ps = 16 * njf, gs = nzs;
    const int kr = ((n * N + HESS_KC - 1) / HESS_KC) * HESS_KC;
    std::vector<void*> tmp;
    auto cleanup = [&]() { for (void* p : tmp) (void)hipFree(p); };
#define DTRY(call)                                                                           \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); cleanup(); return -3; } \
    } while (0)
    auto dnew = [&](double** p, size_t cnt) -> hipError_t {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(p), cnt * sizeof(double) + 64);
        if (e == hipSuccess) { tmp.push_back(*p); e = hipMemsetAsync(*p, 0, cnt * sizeof(double), stream); }
        return e;
    };
    double *dA, *dB, *dQ, *dR, *dS, *dP, *dPhi, *dGk, *dGam, *dW, *dWP, *dH, *dF, *dHs, *dFs, *dVs, *dMinv;
    int* dFlag;
    DTRY(dnew(&dA, (size_t)n * n)); DTRY(dnew(&dB, (size_t)n * m)); DTRY(dnew(&dQ, (size_t)n * n));
    DTRY(dnew(&dR, (size_t)m * m)); DTRY(dnew(&dS, (size_t)m * m)); DTRY(dnew(&dP, (size_t)n * n));
    DTRY(dnew(&dPhi, (size_t)N * n * n)); DTRY(dnew(&dGk, (size_t)N * n * m));
    DTRY(dnew(&dGam, (size_t)kr * gs)); DTRY(dnew(&dW, (size_t)kr * gs)); DTRY(dnew(&dWP, (size_t)kr * ps));
    DTRY(dnew(&dH, (size_t)nz * nz)); DTRY(dnew(&dF, (size_t)nz * n));
    DTRY(dnew(&dHs, (size_t)nz * nzs)); DTRY(dnew(&dFs, (size_t)n * nzs)); DTRY(dnew(&dVs, (size_t)n * nzs)); DTRY(dnew(&dMinv, (size_t)nz * nzs));
    {
        double* f = nullptr;
        DTRY(dnew(&f, 8));
        dFlag = reinterpret_cast<int*>(f);
    }
    DTRY(hipMemcpyAsync(dA, A.data(), A.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    DTRY(hipMemcpyAsync(dB, B.data(), B.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    DTRY(hipMemcpyAsync(dQ, Q.data(), Q.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    DTRY(hipMemcpyAsync(dR, R.data(), R.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    DTRY(hipMemcpyAsync(dS, S.data(), S.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    DTRY(hipMemcpyAsync(dP, P.data(), P.size() * sizeof(double), hipMemcpyHostToDevice, stream));

    hipLaunchKernelGGL(k_design_blocks, dim3(1), dim3(256), (size_t)(3 * n * n + n * m) * sizeof(double), stream, n, m, N,
                       dA, dB, dPhi, dGk, DesignStrides());
    DTRY(hipGetLastError());
    hipLaunchKernelGGL(k_design_gamma, dim3(N), dim3(256), 0, stream, n, m, N, dQ, dP, dPhi, dGk, dGam, dW, dWP, gs, ps, DesignStrides());
    DTRY(hipGetLastError());
    HessParams hp;
    hp.n = n; hp.m = m; hp.N = N; hp.nz = nz; hp.nrb = nrb; hp.njf = njf; hp.kr = kr;
    hp.Gam = dGam; hp.W = dW; hp.WP = dWP; hp.gs = gs; hp.ps = ps; hp.R = dR; hp.S = dS; hp.useR = useR; hp.useS = useS;
    hp.H = dH; hp.F = dF;
    const size_t hess_lds = (size_t)HESS_KC * (16 + (gs + 16) + (ps + 16)) * sizeof(double);
    hipLaunchKernelGGL(k_design_hessian, dim3(nrb), dim3(64 * (nrb + njf)), hess_lds, stream, hp);
    DTRY(hipGetLastError());
    hipLaunchKernelGGL(k_design_scale, dim3(1), dim3(256), 0, stream, nz, nzs, n, dH, dF, dD, dHs, dFs, dFlag, DesignStrides());
    DTRY(hipGetLastError());
    const size_t inv_lds = ((size_t)nz * (nz + 1) + nz + 2) * sizeof(double);
    DTRY(ensure_dyn_lds(reinterpret_cast<const void*>(k_design_inverse_chol), (size_t)(inv_lds)));
    // G = H'^-1 first: the stiffness profile of the ADMM penalty (rho_i = rho / G_ii) is read off its diagonal
    hipLaunchKernelGGL(k_design_inverse_chol, dim3(1), dim3(512), inv_lds, stream, nz, nzs, dHs, 0.0, (const double*)nullptr, dG, dFlag, 0L, 0L, 0L, 0L);
    DTRY(hipGetLastError());
    hipLaunchKernelGGL(k_design_rho, dim3(1), dim3(256), 0, stream, nz, nzs, rho_mode, rho, dG, dRho, 0L, 0L);
    hipLaunchKernelGGL(k_design_inverse_chol, dim3(1), dim3(512), inv_lds, stream, nz, nzs, dHs, sigma, (const double*)dRho, dMinv, dFlag, 0L, 0L, 0L, 0L);
    DTRY(hipGetLastError());
    hipLaunchKernelGGL(k_pack_frags, dim3(32), dim3(256), 0, stream, dMinv, nz, nz, nzs, nrb, ks, dMinvFrag);
    hipLaunchKernelGGL(k_neg_gm, dim3(32), dim3(256), 0, stream, nz, nzs, n, nzs, dG, dFs, dVs, 0L, 0L);
    hipLaunchKernelGGL(k_pack_frags, dim3(32), dim3(256), 0, stream, dVs, nz, n, nzs, nrb, ksf, dVFrag);
    if (dVsOut) DTRY(hipMemcpyAsync(dVsOut, dVs, (size_t)n * nzs * sizeof(double), hipMemcpyDeviceToDevice, stream));   // (plain V: the s0 table of the state-row finish)
    if (dPlainOut) {
        const size_t mm = (size_t)nz * nzs, fv = (size_t)n * nzs;
        DTRY(hipMemcpyAsync(dPlainOut, dMinv, mm * sizeof(double), hipMemcpyDeviceToDevice, stream));
        DTRY(hipMemcpyAsync(dPlainOut + mm, dHs, mm * sizeof(double), hipMemcpyDeviceToDevice, stream));
        DTRY(hipMemcpyAsync(dPlainOut + 2 * mm, dFs, fv * sizeof(double), hipMemcpyDeviceToDevice, stream));
        DTRY(hipMemcpyAsync(dPlainOut + 2 * mm + fv, dVs, fv * sizeof(double), hipMemcpyDeviceToDevice, stream));
    }
    hipLaunchKernelGGL(k_pack_frags, dim3(32), dim3(256), 0, stream, dHs, nz, nz, nzs, nrb, ks, dHFrag);
    hipLaunchKernelGGL(k_pack_frags, dim3(32), dim3(256), 0, stream, dFs, nz, n, nzs, nrb, ksf, dFFrag);
    DTRY(hipGetLastError());
    if (!rowsel.empty()) {  // state rows: Ghat = A G A' in constraint space
        const int mc = (int)rowsel.size();
        double *dCp, *dCG;
        int* dSel;
        DTRY(dnew(&dCp, (size_t)mc * nzs)); DTRY(dnew(&dCG, (size_t)mc * nzs));
        {
            double* tmpi = nullptr;
            DTRY(dnew(&tmpi, (size_t)(mc + 1) / 2 + 1));
            dSel = reinterpret_cast<int*>(tmpi);
        }
        DTRY(hipMemcpyAsync(dSel, rowsel.data(), mc * sizeof(int), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_build_cprime, dim3(64), dim3(256), 0, stream, mc, nz, nzs, gs, dSel, dGam, dD, dCp);
        hipLaunchKernelGGL(k_cg, dim3(256), dim3(256), 0, stream, mc, nz, nzs, dCp, dG, dCG);
        hipLaunchKernelGGL(k_ghat, dim3(512), dim3(256), 0, stream, nz, mc, nzs, Rs, dG, dCp, dCG, dGhat, dGnorm);
        DTRY(hipGetLastError());
    }
    hH.assign((size_t)nz * nz, 0.0); hF.assign((size_t)nz * n, 0.0); hd.assign((size_t)nzs, 0.0);
    int flag = 0;
    DTRY(hipMemcpyAsync(hH.data(), dH, hH.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
    DTRY(hipMemcpyAsync(hF.data(), dF, hF.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
    DTRY(hipMemcpyAsync(hd.data(), dD, hd.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
    DTRY(hipMemcpyAsync(&flag, dFlag, sizeof(int), hipMemcpyDeviceToHost, stream));
    DTRY(hipStreamSynchronize(stream));
#undef DTRY
    cleanup();
    if (flag != 0) {
        err = flag == 1 ? "design: condensed Hessian has a non-positive diagonal (R = 0 with an input that does not reach the cost?)"
                        : "design: Cholesky pivot not positive (Hessian not positive definite)";
        return -6;
    }
    return 0;
}

}  // namespace almpc
