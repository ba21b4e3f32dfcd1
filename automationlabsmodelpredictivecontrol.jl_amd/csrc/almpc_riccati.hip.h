// almpc_riccati.hip.h -- the structured (non-condensed) solve: SURVEY.md section 8f rank 4, second half.
//
// The reference keeps the multiple-shooting form of the QP (variables x, u per stage, dynamics as constraints:
// src/sub/model_modeler_implementation/linear/mpc_modeler_implementation_linear.jl:48-60) because condensing is the weak spot of a
// dense solver: the condensed Hessian of an open-loop unstable model over a long horizon loses definiteness in FP64 and its size is
// (m N)^2.  k_riccati solves that form directly: a primal active-set method on the input box whose equality-constrained
// subproblems (inputs of the working set held at their bounds) are a backward Riccati recursion plus a forward rollout,
//     Lam = R_ff + B_f' P+ B_f,  K = Lam^-1 B_f' P+ A,  kff = Lam^-1 (B_f'(P+ d + p+) + R_fF b_F),  d = B_F b_F,
//     P   = Q + A' P+ A - (A' P+ B_f) K,   p = A'(P+ d + p+) - (A' P+ B_f) kff,          v_f = -K e - kff,
// O(N (n^3 + n^2 m)) per working-set change, stable for any spectral radius, no m N limit.  The multipliers of the bounds come from
// the adjoint recursion along the trajectory (lam_N = P e_N, lam_k = Q e_k + A' lam_{k+1}, mu_k = 2 (R v_k + B' lam_{k+1})).
// Cost: the reference's (src/sub/design_mpc.jl:405-468, S = 0): e_N' P e_N + sum_{k=1..N-1} e_k' Q e_k + sum_k v_k' R v_k, no 1/2.
// Oracle: oracle/mpc_oracle.py::riccati_active_set (same decisions: ratio test ties -> smallest stage, then smallest input).
//
// One wave per instance; every matrix of the recursion lives in the wave's slice of LDS (column-major, as the ABI's), gains K_k,
// kff_k go to a per-instance scratch in global memory (read back one stage ahead in the forward pass).  Used (i) as the solver of
// handles created with ALMPC_FLAG_STRUCTURED (any m N <= 1024, n <= 32, m <= 16) and (ii) as the fallback for instances the
// condensed path leaves unsolved (per-instance models with unstable linearisations).  Correctness path: not tuned.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace almpc {

struct RiccatiParams {
    int n, m, N, batch;
    const double* A;  long A_stride;   // n x n column-major (stride 0: shared by the batch)
    const double* B;  long B_stride;   // n x m
    const double* Q;                   // n x n (shared), symmetric
    const double* R;                   // m x m (shared), symmetric (zeros if the reference's branch rule drops the input term)
    const double* P;  long P_stride;   // n x n terminal weight
    const double* umin; const double* umax;   // [m]
    const double* uref; long uref_stride;     // [N][m]
    const double* xref; long xref_stride;     // [N+1][n]
    const double* x0;                          // [batch][x0_stride >= n]
    long x0_stride;
    const double* uguess;                      // [batch][N][m] start (e.g. the condensed path's result) or null: clipped LQR
    int filter;                                // 1: only instances with status != 0 (fallback after the condensed path);
                                               // 2: only instances with flag != 0 or status != 0, flag cleared when solved (SQP loop)
    // time-varying stage models (the QP of one SQP iteration, almpc_sqp.hip.h): A_k = A + inst * A_stride + k * A_kstride etc.;
    // c [N][n] defects, ebar [N][n]: the cost is on e_{k+1} + ebar_k (state error of the linearisation point), qu [N][m]: linear
    // input cost (J += 2 qu'v); e_0 = 0 when x0 is null.  All nullable / zero for the time-invariant MPC problem.
    long A_kstride, B_kstride;
    const double* c;    long c_stride;
    const double* ebar; long ebar_stride;
    const double* qu;   long qu_stride; double qu_scale;   // (J += 2 qu_scale qu'v: 0.5 for a plain gradient vector)
    int* flag;                                 // [batch] (filter 2)
    int v_only;                                // 1: write e_u (= v) and status / piters only (the SQP update does the rest)
    double* Kst;                               // scratch [batch][N][m*n + m]: K_k | kff_k
    double* Pst;                               // scratch [batch][N][n*n + n]: P_k | p_k of the last sweep (k = 1..N-1), or null: a working-set
                                               // change at stage k leaves the recursion above k untouched, so the next backward sweep
                                               // restarts from P_{k+1}, p_{k+1} instead of from the terminal weight
    double* x; double* ex; double* u; double* eu;   // results, layouts of almpc_get_results
    int32_t* status; int32_t* piters;
    int max_iter;
    double tol;
    int lds_per_wave;                          // doubles
    const int* gate = nullptr; int gate_val = 0;   // the launch does nothing unless *gate == gate_val (SdualParams::gate)
};

__host__ __device__ inline int riccati_lds_doubles(int n, int m, int N) {
    // A, Q, Pn, M1: 4 n^2 | B, PB, BPA, K: 4 n m | R, Lam, Li: 3 m^2 | vectors: 8 max(n, m) | v, vs: 2 N m | bval: N m |
    // trajectory (N+1) n | wset: N ints
    const int mx = n > m ? n : m;
    return (4 * n * n + 4 * n * m + 3 * m * m + 8 * mx + 3 * N * m + (N + 1) * n + (N + 1) / 2 + 3) & ~1;
}

__device__ __forceinline__ void rw_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// out(r x c) = op(X)(r x k) * Y(k x c), column-major, by one wave; TX: X is stored k x r and used transposed
template <bool TX>
__device__ __forceinline__ void wv_matmul(double* out, const double* X, const double* Y, int r, int k, int c, int lane) {
    for (int t = lane; t < r * c; t += 64) {
        const int i = t % r, j = t / r;
        double acc = 0.0;
        for (int l = 0; l < k; ++l) acc += (TX ? X[l + i * k] : X[i + l * r]) * Y[l + j * k];
        out[t] = acc;
    }
}

__device__ __forceinline__ double wave_min_d(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
    for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(v, o); v = w < v ? w : v; }
    return v;
}

// Start of a structured solve from another solve's inputs: dst[i][k][a] = src[i][k + shift][a] for the stages the source has
// (shift = 1: the receding-horizon shift of the previous step, the last stage repeated), the input reference beyond them.  The
// source may have a shorter horizon (horizon continuation: the condensed path at N = 30 hands the stage-wise path at N = 50 its
// working set -- the active bounds of an MPC problem with the DARE terminal weight sit in the early stages).
inline __global__ __launch_bounds__(256) void k_guess_from_inputs(int batch, int m, int Nsrc, int N, int shift, const double* usrc, const double* uref,
                                                           long uref_stride, double* dst) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)batch * N * m) return;
    const int a = (int)(t % m), k = (int)((t / m) % N);
    const long i = t / ((long)m * N);
    int ks = k + shift;
    if (shift && ks >= Nsrc) ks = Nsrc - 1;
    dst[t] = (ks < Nsrc) ? usrc[(i * Nsrc + ks) * m + a] : uref[i * uref_stride + (long)k * m + a];
}

constexpr int RICCATI_WAVES = 4;
constexpr int RICCATI_CLIP_START = 4;   // see the start of the active-set loop

// <NC, MC>: state / input dimensions known at compile time (0, 0: read from the parameters).  Every product of a stage is a short
// loop over n or m with two LDS reads per term; with run-time bounds each term is its own LDS round trip (measured: 29.6 k cycles
// per stage of a sweep at n 12, m 4, a quarter of it in the forward rollout's two 12-term loops), with compile-time bounds the
// loops are unrolled and their reads go out together.
template <int NC, int MC>
__global__ __launch_bounds__(64 * RICCATI_WAVES) void k_riccati_t(RiccatiParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (p.gate && __builtin_nontemporal_load(p.gate) != p.gate_val) return;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = NC ? NC : p.n, m = MC ? MC : p.m, N = p.N, nm = n * m, nn = n * n, mm = m * m;
    const int mx = n > m ? n : m;
    double* L = smem + (size_t)wv * p.lds_per_wave;
    double* As = L;            double* Qs = As + nn;     double* Pn = Qs + nn;    double* M1 = Pn + nn;
    double* Bs = M1 + nn;      double* PB = Bs + nm;     double* BPA = PB + nm;   double* Ks = BPA + nm;
    double* Rs = Ks + nm;      double* Lam = Rs + mm;    double* Li = Lam + mm;
    double* vec = Li + mm;     // 8 vectors of length mx
    double* dv = vec;          double* gv = vec + mx;    double* pn = vec + 2 * mx;  double* hv = vec + 3 * mx;
    double* kffv = vec + 4 * mx; double* ev = vec + 5 * mx; double* lamv = vec + 6 * mx; double* btv = vec + 7 * mx;
    double* vcur = vec + 8 * mx;      // [N][m] current feasible point
    double* vstar = vcur + N * m;     // [N][m] minimiser of the current face
    double* bval = vstar + N * m;     // [N][m] bound a fixed input sits on
    double* etraj = bval + N * m;     // [N+1][n] e_x of the last rollout
    uint32_t* wset = reinterpret_cast<uint32_t*>(etraj + (N + 1) * n);   // [N] bit a: input a of the stage is fixed; bit 16 + a: at its upper bound

    const int wpb = (int)(blockDim.x >> 6);   // waves per workgroup: as many (<= RICCATI_WAVES) as have room for their LDS slice
    const int nwaves = gridDim.x * wpb;
    for (int inst = blockIdx.x * wpb + wv; inst < p.batch; inst += nwaves) {
        if (p.filter == 1 && p.status[inst] == 0) continue;
        if (p.filter == 2 && p.status[inst] == 0 && p.flag[inst] == 0) continue;
        const double* Ag = p.A + (size_t)inst * p.A_stride;
        const double* Bg = p.B + (size_t)inst * p.B_stride;
        const double* Pg = p.P + (size_t)inst * p.P_stride;
        const double* urg = p.uref + (size_t)inst * p.uref_stride;
        const double* xrg = p.xref + (size_t)inst * p.xref_stride;
        double* Kg = p.Kst + (size_t)inst * N * (nm + m);
        double* exg = etraj;   // the trajectory stays in LDS (re-read across lanes by the adjoint pass); copied out at the end
        const bool ltv = p.A_kstride != 0;
        const double* cg = p.c ? p.c + (size_t)inst * p.c_stride : nullptr;
        const double* ebg = p.ebar ? p.ebar + (size_t)inst * p.ebar_stride : nullptr;
        const double* qug = p.qu ? p.qu + (size_t)inst * p.qu_stride : nullptr;
        for (int t = lane; t < nn; t += 64) { As[t] = Ag[t]; Qs[t] = p.Q[t]; }
        for (int t = lane; t < nm; t += 64) Bs[t] = Bg[t];
        for (int t = lane; t < mm; t += 64) Rs[t] = p.R[t];
        rw_fence();
        auto load_stage = [&](int k) {   // time-varying models: (A_k, B_k) into the LDS slots of the stage loops
            if (!ltv) return;
            rw_fence();
            for (int t = lane; t < nn; t += 64) As[t] = Ag[(size_t)k * p.A_kstride + t];
            for (int t = lane; t < nm; t += 64) Bs[t] = Bg[(size_t)k * p.B_kstride + t];
            rw_fence();
        };
        auto lo_of = [&](int k, int a) { return p.umin[a] - urg[k * m + a]; };
        auto hi_of = [&](int k, int a) { return p.umax[a] - urg[k * m + a]; };

        // ---- backward Riccati sweep for the current working set, then the forward rollout: vstar, trajectory in exg
        double* Pstg = p.Pst ? p.Pst + (size_t)inst * N * (nn + n) : nullptr;
#ifdef ALMPC_STAMPS
        long long rk_acc[8], rk_t = 0;
#define RK_T0() do { __builtin_amdgcn_sched_barrier(0); rk_t = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define RK_ACC(S) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = __builtin_readcyclecounter(); rk_acc[S] += t_ - rk_t; rk_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RK_T0()
#define RK_ACC(S)
#endif
        auto sweep = [&](int kstart) {   // kstart: the highest stage whose working set changed since the last sweep (N - 1: all)
            if (!Pstg || kstart >= N - 1) {
                kstart = N - 1;
                for (int t = lane; t < nn; t += 64) Pn[t] = 0.5 * (Pg[t] + Pg[(t % n) * n + t / n]);
                rw_fence();
                for (int i = lane; i < n; i += 64) {   // p_N = P ebar_{N-1}: the terminal cost is on e_N + ebar_{N-1}
                    double s = 0.0;
                    if (ebg) for (int j = 0; j < n; ++j) s += Pn[i + j * n] * ebg[(size_t)(N - 1) * n + j];
                    pn[i] = s;
                }
            } else {   // P_{kstart+1}, p_{kstart+1} of the previous sweep (stored by other lanes of this wave: device-coherent loads)
                const double* src = Pstg + (size_t)(kstart + 1) * (nn + n);
                for (int t = lane; t < nn; t += 64) Pn[t] = __hip_atomic_load(src + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int t = lane; t < n; t += 64) pn[t] = __hip_atomic_load(src + nn + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            rw_fence();
            for (int k = kstart; k >= 0; --k) {
                const uint32_t wk = wset[k];
                RK_T0();
                load_stage(k);
                for (int a = lane; a < m; a += 64) btv[a] = ((wk >> a) & 1u) ? bval[k * m + a] : 0.0;
                rw_fence();
                for (int i = lane; i < n; i += 64) {   // d = B bt + c_k
                    double s = cg ? cg[(size_t)k * n + i] : 0.0;
                    for (int a = 0; a < m; ++a) s += Bs[i + a * n] * btv[a];
                    dv[i] = s;
                }
                wv_matmul<false>(M1, Pn, As, n, n, n, lane);   // M1 = P+ A
                wv_matmul<false>(PB, Pn, Bs, n, n, m, lane);   // PB = P+ B
                rw_fence();
                RK_ACC(0);
                for (int i = lane; i < n; i += 64) {   // g = P+ d + p+
                    double s = pn[i];
                    for (int j = 0; j < n; ++j) s += Pn[i + j * n] * dv[j];
                    gv[i] = s;
                }
                wv_matmul<true>(BPA, Bs, M1, m, n, n, lane);   // B' P+ A   (m x n)
                wv_matmul<true>(Lam, Bs, PB, m, n, m, lane);   // B' P+ B   (m x m)
                rw_fence();
                RK_ACC(1);
                for (int t = lane; t < mm; t += 64) {          // Lam = R + B'P+B, identity on the fixed inputs
                    const int i = t % m, j = t / m;
                    const bool fi = (wk >> i) & 1u, fj = (wk >> j) & 1u;
                    Lam[t] = (fi || fj) ? (i == j ? 1.0 : 0.0) : Lam[t] + Rs[t];
                }
                for (int a = lane; a < m; a += 64) {           // h = B' g + R bt + qu_k, zero on the fixed inputs
                    double s = qug ? p.qu_scale * qug[(size_t)k * m + a] : 0.0;
                    for (int i = 0; i < n; ++i) s += Bs[i + a * n] * gv[i];
                    for (int b = 0; b < m; ++b) s += Rs[a + b * m] * btv[b];
                    hv[a] = ((wk >> a) & 1u) ? 0.0 : s;
                }
                for (int t = lane; t < nm; t += 64)            // rows of the fixed inputs of B'P+A vanish
                    if ((wk >> (t % m)) & 1u) BPA[t] = 0.0;
                rw_fence();
                RK_ACC(2);
                // Li = Lam^-1 by in-place Gauss-Jordan (SPD: no pivoting)
                for (int t = lane; t < mm; t += 64) Li[t] = Lam[t];
                rw_fence();
                for (int pv = 0; pv < m; ++pv) {
                    double nv[4];
                    int cnt = 0;
                    const double ip = 1.0 / Li[pv + pv * m];
                    for (int t = lane; t < mm; t += 64) {
                        const int i = t % m, j = t / m;
                        const double lip = Li[i + pv * m], lpj = Li[pv + j * m];
                        nv[cnt++] = (i == pv) ? ((j == pv) ? ip : lpj * ip) : ((j == pv) ? -lip * ip : Li[t] - lip * lpj * ip);
                    }
                    rw_fence();
                    cnt = 0;
                    for (int t = lane; t < mm; t += 64) Li[t] = nv[cnt++];
                    rw_fence();
                }
                RK_ACC(3);
                wv_matmul<false>(Ks, Li, BPA, m, m, n, lane);  // K = Lam^-1 B'P+A
                for (int a = lane; a < m; a += 64) {
                    double s = 0.0;
                    for (int b = 0; b < m; ++b) s += Li[a + b * m] * hv[b];
                    kffv[a] = s;
                }
                rw_fence();
                for (int t = lane; t < nm; t += 64) Kg[(size_t)k * (nm + m) + t] = Ks[t];
                for (int a = lane; a < m; a += 64) Kg[(size_t)k * (nm + m) + nm + a] = kffv[a];
                RK_ACC(4);
                if (k > 0) {
                    // P = Q + A' M1 - BPA' K (symmetrised), p = A' g - BPA' kff
                    double pnew[16];   // n*n <= 1024 -> at most 16 elements per lane
                    int cnt = 0;
                    for (int t = lane; t < nn; t += 64) {
                        const int i = t % n, j = t / n;
                        double s = Qs[t];
                        for (int l = 0; l < n; ++l) s += As[l + i * n] * M1[l + j * n];
                        for (int a = 0; a < m; ++a) s -= BPA[a + i * m] * Ks[a + j * m];
                        pnew[cnt++] = s;
                    }
                    double pv2 = 0.0;
                    if (lane < n) {
                        double s = 0.0;
                        if (ebg) for (int j = 0; j < n; ++j) s += Qs[lane + j * n] * ebg[(size_t)(k - 1) * n + j];   // + Q ebar_{k-1}: cost on e_k + ebar_{k-1}
                        for (int l = 0; l < n; ++l) s += As[l + lane * n] * gv[l];
                        for (int a = 0; a < m; ++a) s -= BPA[a + lane * m] * kffv[a];
                        pv2 = s;
                    }
                    rw_fence();
                    cnt = 0;
                    for (int t = lane; t < nn; t += 64) M1[t] = pnew[cnt++];   // M1 is free now: holds the unsymmetrised P
                    if (lane < n) pn[lane] = pv2;
                    rw_fence();
                    for (int t = lane; t < nn; t += 64) {
                        const double v = 0.5 * (M1[t] + M1[(t % n) * n + t / n]);
                        Pn[t] = v;
                        if (Pstg) Pstg[(size_t)k * (nn + n) + t] = v;
                    }
                    if (Pstg && lane < n) Pstg[(size_t)k * (nn + n) + nn + lane] = pv2;
                    rw_fence();
                }
                RK_ACC(5);
            }
            RK_T0();
            // forward: e_0 = x0 - x_ref[:,0];  v*_k = fixed ? bound : -K_k e_k - kff_k;  e_{k+1} = A e_k + B v*_k
            // (the gains were stored by other lanes of this wave: made visible at device scope, read back past the L1 and one stage ahead)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_wave_barrier();
            if (lane < n) { const double e0 = p.x0 ? p.x0[(size_t)inst * p.x0_stride + lane] - xrg[lane] : 0.0; ev[lane] = e0; exg[lane] = e0; }
            double kreg[9];   // (n m + m) / 64 <= 9 for n <= 32, m <= 16
            auto kload = [&](int k) {
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    const int t = lane + 64 * c;
                    kreg[c] = (t < nm + m) ? __hip_atomic_load(Kg + (size_t)k * (nm + m) + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                }
            };
            kload(0);
            rw_fence();
            for (int k = 0; k < N; ++k) {
                const uint32_t wk = wset[k];
#pragma unroll
                for (int c = 0; c < 9; ++c) {
                    const int t = lane + 64 * c;
                    if (t < nm) Ks[t] = kreg[c];
                    else if (t < nm + m) kffv[t - nm] = kreg[c];
                }
                if (k + 1 < N) kload(k + 1);
                load_stage(k);
                rw_fence();
                for (int a = lane; a < m; a += 64) {
                    double s = -kffv[a];
                    for (int j = 0; j < n; ++j) s -= Ks[a + j * m] * ev[j];
                    vstar[k * m + a] = ((wk >> a) & 1u) ? bval[k * m + a] : s;
                }
                rw_fence();
                double en = 0.0;
                if (lane < n) {
                    en = cg ? cg[(size_t)k * n + lane] : 0.0;
                    for (int j = 0; j < n; ++j) en += As[lane + j * n] * ev[j];
                    for (int a = 0; a < m; ++a) en += Bs[lane + a * n] * vstar[k * m + a];
                }
                rw_fence();
                if (lane < n) { ev[lane] = en; exg[(size_t)(k + 1) * n + lane] = en; }
                rw_fence();
            }
            RK_ACC(6);
        };
        // trajectory of vcur (after a partial step the stored one is vstar's)
        auto rollout_cur = [&]() {
            if (lane < n) { const double e0 = p.x0 ? p.x0[(size_t)inst * p.x0_stride + lane] - xrg[lane] : 0.0; ev[lane] = e0; exg[lane] = e0; }
            rw_fence();
            for (int k = 0; k < N; ++k) {
                load_stage(k);
                double en = 0.0;
                if (lane < n) {
                    en = cg ? cg[(size_t)k * n + lane] : 0.0;
                    for (int j = 0; j < n; ++j) en += As[lane + j * n] * ev[j];
                    for (int a = 0; a < m; ++a) en += Bs[lane + a * n] * vcur[k * m + a];
                }
                rw_fence();
                if (lane < n) { ev[lane] = en; exg[(size_t)(k + 1) * n + lane] = en; }
                rw_fence();
            }
        };

        // ---- start: the given guess, or the clipped unconstrained (LQR) solution; inputs on a bound enter the working set
        for (int k = lane; k < N; k += 64) wset[k] = 0u;
        for (int t = lane; t < N * m; t += 64) bval[t] = 0.0;
        rw_fence();
        if (p.uguess) {
            for (int t = lane; t < N * m; t += 64) vstar[t] = p.uguess[(size_t)inst * N * m + t] - urg[t];
            rw_fence();
        } else {
            sweep(N - 1);
            // the clipped unconstrained solution is a good start only when it clips few inputs: with many, most of the clipped rows are
            // not active at the optimum and each costs a sweep to remove (measured on the quadrotor at N = 50: 339 changes against 60
            // from the interior).  More than RICCATI_CLIP_START rows clipped: start from the reference input (v = 0) instead.
            int nclip = 0;
            for (int t = lane; t < N * m; t += 64) {
                const int k = t / m, a = t % m;
                nclip += (vstar[t] > hi_of(k, a) || vstar[t] < lo_of(k, a)) ? 1 : 0;
            }
            for (int o = 32; o > 0; o >>= 1) nclip += __shfl_xor(nclip, o);
            if (nclip > RICCATI_CLIP_START) {
                rw_fence();
                for (int t = lane; t < N * m; t += 64) vstar[t] = 0.0;
                rw_fence();
            }
        }
        for (int t = lane; t < N * m; t += 64) {
            const int k = t / m, a = t % m;
            const double lo = lo_of(k, a), hi = hi_of(k, a);
            double v = vstar[t];
            v = v != v ? 0.0 : v;   // (a non-finite guess starts from the reference)
            v = fmin(fmax(v, lo), hi);
            vcur[t] = v;
            if (v >= hi) { atomicOr(&wset[k], (1u << a) | (1u << (16 + a))); bval[t] = hi; }
            else if (v <= lo) { atomicOr(&wset[k], 1u << a); bval[t] = lo; }
        }
        rw_fence();

        int it = 0, fin = 1;
        bool bad = false;
#ifdef ALMPC_STAMPS
        for (int c_ = 0; c_ < 8; ++c_) rk_acc[c_] = 0;
#endif
        int kchg = N - 1;   // highest stage whose working set changed since the last sweep
        while (it < p.max_iter) {
            ++it;
            sweep(kchg);
            // ---- ratio test over the free inputs of all stages: first bound hit on the way from vcur to vstar
            double rmin = __builtin_inf();
            int tmin = 0x7fffffff;
            double chk = 0.0;
            for (int t = lane; t < N * m; t += 64) {
                const int k = t / m, a = t % m;
                const double vs_ = vstar[t];
                chk += vs_ - vs_;   // NaN / inf detector
                if ((wset[k] >> a) & 1u) continue;
                const double lo = lo_of(k, a), hi = hi_of(k, a), v = vcur[t];
                double rr = __builtin_inf();
                if (vs_ > hi) rr = (hi - v) / (vs_ - v);
                else if (vs_ < lo) rr = (lo - v) / (vs_ - v);
                if (rr < rmin) { rmin = rr; tmin = t; }
            }
            if (__any(chk != 0.0)) { bad = true; break; }
            const double rall = wave_min_d(rmin);
            if (rall < 1.0) {
                const int tsel = wave_min_i(rmin == rall ? tmin : 0x7fffffff);
                const double tt = fmax(rall, 0.0);
                for (int t = lane; t < N * m; t += 64) {
                    const int k = t / m, a = t % m;
                    if (!((wset[k] >> a) & 1u)) vcur[t] += tt * (vstar[t] - vcur[t]);
                }
                rw_fence();
                if (lane == 0) {
                    const int k = tsel / m, a = tsel % m;
                    const bool up = vstar[tsel] > hi_of(k, a);
                    const double b = up ? hi_of(k, a) : lo_of(k, a);
                    vcur[tsel] = b; bval[tsel] = b;
                    wset[k] |= (1u << a) | (up ? (1u << (16 + a)) : 0u);
                }
                kchg = tsel / m;
                rw_fence();
                continue;
            }
            for (int t = lane; t < N * m; t += 64) vcur[t] = vstar[t];
            rw_fence();
            // ---- multipliers of the fixed inputs by the adjoint recursion (the stored trajectory is vstar's = vcur's)
            if (lane < n) {   // lam_N = P (e_N + ebar_{N-1})
                double s = 0.0;
                for (int j = 0; j < n; ++j)
                    s += 0.5 * (Pg[lane + j * n] + Pg[j + lane * n]) * (exg[(size_t)N * n + j] + (ebg ? ebg[(size_t)(N - 1) * n + j] : 0.0));
                lamv[lane] = s;
            }
            rw_fence();
            double vmax = -__builtin_inf(), mumax = 0.0;
            int tv = 0x7fffffff;
            for (int k = N - 1; k >= 0; --k) {
                const uint32_t wk = wset[k];
                load_stage(k);
                if (lane < m) {
                    double s = qug ? p.qu_scale * qug[(size_t)k * m + lane] : 0.0;
                    for (int b = 0; b < m; ++b) s += Rs[lane + b * m] * vcur[k * m + b];
                    for (int i = 0; i < n; ++i) s += Bs[i + lane * n] * lamv[i];
                    const double mu = 2.0 * s;
                    if ((wk >> lane) & 1u) {
                        const double viol = ((wk >> (16 + lane)) & 1u) ? mu : -mu;   // upper bound needs mu <= 0, lower bound mu >= 0
                        mumax = fmax(mumax, fabs(mu));
                        if (viol > vmax || (viol == vmax && k * m + lane < tv)) { vmax = viol; tv = k * m + lane; }
                    }
                }
                double ln = 0.0;
                if (lane < n && k > 0) {   // lam_k = Q (e_k + ebar_{k-1}) + A_k' lam_{k+1}  (stage 0 carries no cost)
                    for (int j = 0; j < n; ++j)
                        ln += Qs[lane + j * n] * (exg[(size_t)k * n + j] + (ebg ? ebg[(size_t)(k - 1) * n + j] : 0.0)) + As[j + lane * n] * lamv[j];
                }
                rw_fence();
                if (lane < n) lamv[lane] = ln;
                rw_fence();
            }
            const double vall = wave_max_d(vmax), mall = wave_max_d(mumax);
            if (!(vall > p.tol * fmax(1.0, mall))) { fin = 0; break; }
            const int tsel = wave_min_i(vmax == vall ? tv : 0x7fffffff);
            if (lane == 0) {
                const int k = tsel / m, a = tsel % m;
                wset[k] &= ~((1u << a) | (1u << (16 + a)));
                bval[tsel] = 0.0;
            }
            kchg = tsel / m;
            rw_fence();
        }
#ifdef ALMPC_STAMPS
        if (g_stamps && lane == 0) {
            for (int c_ = 0; c_ < 7; ++c_) g_stamps[(size_t)inst * 16 + c_] = rk_acc[c_];
            g_stamps[(size_t)inst * 16 + 8] = it;
        }
#endif
        // ---- outputs: u = v + u_ref (clamped to the box), e_u, x = e_x + x_ref, e_x (trajectory of the final point)
        if (p.v_only) {   // SQP loop: the QP's solution v (= e_u of the handle) is all the update kernel needs
            for (int t = lane; t < N * m; t += 64) {
                const int a = t % m;
                const double ur = urg[t];
                p.eu[(size_t)inst * N * m + t] = bad ? vcur[t] : fmin(fmax(vcur[t] + ur, p.umin[a]), p.umax[a]) - ur;
            }
            if (lane == 0) {
                p.status[inst] = bad ? 2 : (fin == 0 ? 0 : 1);
                p.piters[inst] = it;
                if (p.filter == 2 && !bad && fin == 0) p.flag[inst] = 0;
            }
            rw_fence();
            continue;
        }
        if (fin != 0 && !bad) rollout_cur();
        for (int t = lane; t < N * m; t += 64) {
            const int a = t % m;
            const double ur = urg[t];
            double uu = fmin(fmax(vcur[t] + ur, p.umin[a]), p.umax[a]);
            if (bad) uu = vcur[t] + ur;
            p.u[(size_t)inst * N * m + t] = uu;
            p.eu[(size_t)inst * N * m + t] = uu - ur;
        }
        rw_fence();
        for (int t = lane; t < (N + 1) * n; t += 64) {
            const double e = exg[t];
            p.ex[(size_t)inst * (N + 1) * n + t] = e;
            p.x[(size_t)inst * (N + 1) * n + t] = (t < n) ? p.x0[(size_t)inst * p.x0_stride + t] : e + xrg[t];
        }
        if (lane == 0) {
            p.status[inst] = bad ? 2 : (fin == 0 ? 0 : 1);
            p.piters[inst] = it;
        }
        rw_fence();
    }
}

}  // namespace almpc
