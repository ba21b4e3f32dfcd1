// almpc_sdual.hip.h -- the stage-wise (multiple-shooting) QP solved by a DUAL active-set method whose linear algebra is affine
// Riccati sweeps.  SURVEY.md section 8f rank 1 + rank 4: state box on every stage, terminal equality and the input-rate weight S in
// the form the reference itself poses (variables x, u per stage, dynamics as constraints):
//     dynamics / input box / state box     src/sub/model_modeler_implementation/linear/mpc_modeler_implementation_linear.jl:48-78
//     terminal equality e_x[:,N+1] = 0     src/sub/design_mpc.jl:330-331
//     cost, input-rate term on delta_u     src/sub/design_mpc.jl:405-468 (S: :423-446)
// Oracle: oracle/stagewise_oracle.py::solve_stage_dual (same decisions).
//
// Every row of the problem -- an input bound, a state bound, a terminal-equality row -- is ONE COORDINATE of the stage-wise trajectory
// z = (x_0, v_0, x_1, v_1, ..., x_N).  Goldfarb-Idnani in constraint space needs columns of Ghat = A H^-1 A': the response of the
// UNCONSTRAINED problem to a unit linear cost on a coordinate, i.e. one affine backward / forward sweep with the unconstrained feedback
// gains K_k.  The gains do not depend on the working set, so the O(N n^3) Riccati recursion runs once per model (on the host for a
// shared model; k_sgains for models per instance), never per working-set change; a change costs two O(N (n^2 + n m)) sweeps whatever
// the spectral radius of A (the forward sweep is in closed-loop form), plus a bordered update of Sinv = (Ghat_WW)^-1.  With S != 0 the
// stage state is [e_k; v_{k-1}] (nt = n + m).
//
// One wave per instance.  A sweep keeps the stage matrices in registers, lane = row:
//     backward  lane l < nt: p_l = src_x[l] + (Acl' g)_l - (K' src_u)_l,   lane NT + a: kff_a = (Lb g)_a + (Li src_u)_a,   g = p+ (+ P+ c)
//     forward   lane l < nt: x+_l = (Acl x)_l - (Bt kff)_l (+ c_l),        lane NT + a: v_a = -(K x)_a - kff_a
// with the vectors g / x broadcast by wave-uniform LDS reads: one LDS round trip per stage and direction.  The per-stage rows
// [Acl' | -K' | Acl | -Bt] and [Lb | Li | -K | -I] come "sweep-ready" from the gain computation (SdualRecord below); when they do not
// depend on the stage (shared model with the DARE terminal weight, no S) they are loaded once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace almpc {

struct SdualParams {
    int n, nt, m, N, batch;            // n: rows of the stage state that carry bounds (e_k); nt = n (S = 0) or n + m
    const double* rec; long rec_stride, rec_kstride;   // stage records [N][NT + MC rows][2 NT + 2 MC] (rec_stride 0: shared; rec_kstride 0: stage-invariant)
    const double* base; long base_stride;              // [TP] linear cost terms on the coordinates (J += 2 base't), or null: none
    const double* pc; const double* ct; long pc_stride;   // [N][NT] P_{k+1} c_k and c_k (defects of an SQP iteration's QP), or null
    const double* umin; const double* umax;            // [m]
    const double* uref; long uref_stride;              // [N][m]: input bounds are umin - uref_k .. umax - uref_k
    const double* xmin; const double* xmax;            // [n] or null: no state box
    const double* xbref; long xbref_stride;            // [N+1][n]: state bounds are xmin - xbref_k .. xmax - xbref_k
    const double* eqt; long eqt_stride;                // [n] terminal equality target for e_N, or null
    const double* x0; const double* xref; long xref_stride;   // e_0 = x0 - xref_0 (x0 null: e_0 = 0); x = e + xref
    const double* uguess;                              // [batch][N][m] inputs whose bounds seed the working set, or null
    int filter;                                        // 0 all instances; 1 status != 0; 2 status != 0 or flag != 0 (flag cleared when solved)
    int* flag;
    int v_only;                                        // 1: write e_u, status, piters only
    double* x; double* ex; double* u; double* eu;      // results, layouts of almpc_get_results
    int32_t* status; int32_t* piters;
    int32_t* ovf;                                      // [batch] or null: 1 = the working set outgrew wcap (the caller redoes it with more room)
    int only_ovf;                                      // 1: only instances with ovf != 0 (second tier)
    int rows_state;                                    // 1: the problem has state rows (an "infeasible" verdict is meaningful)
    int max_iter;
    double tol;
    int wcap;                                          // working-set capacity (<= 64)
    int lds_per_wave;                                  // doubles
};

__host__ __device__ inline int sdual_tp(int NT, int MC, int N) { return (N * (NT + MC) + NT + 1) & ~1; }
__host__ __device__ inline int sdual_lds_doubles(int NT, int MC, int N, int wcap) {
    // s, w: 2 TP | Sinv: wcap (wcap + 1) | gbuf NT | cbuf, ubuf: 2 wcap
    return (2 * sdual_tp(NT, MC, N) + wcap * (wcap + 1) + NT + 2 * wcap + 4 + 1) & ~1;
}
__host__ __device__ inline int sdual_rec_row(int NT, int MC) { return 2 * NT + 2 * MC; }
__host__ __device__ inline int sdual_rec_stage(int NT, int MC) { return (NT + MC) * sdual_rec_row(NT, MC); }

__device__ __forceinline__ void sd_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double sd_wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int sd_wave_max_i(int v) {
    for (int o = 32; o > 0; o >>= 1) { const int w = __shfl_xor(v, o); v = w > v ? w : v; }
    return v;
}
// (value, index) reductions: larger / smaller value wins, ties -> smaller index
__device__ __forceinline__ void sd_argmax(double& v, int& idx) {
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(idx, o);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}
__device__ __forceinline__ void sd_argmin(double& v, int& idx) {
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(idx, o);
        if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

constexpr int SDUAL_WAVES = 4;

template <int NT, int MC>
__global__ __launch_bounds__(64 * SDUAL_WAVES) void k_sdual(SdualParams p) {
    static_assert((NT % 2) == 0 && (MC % 2) == 0, "even dimensions (16-byte loads)");
    constexpr int SP = NT + MC;          // coordinates per stage: state slot [0, NT), input slot [NT, NT + MC)
    constexpr int RL = 2 * NT + 2 * MC;  // doubles per record row
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = p.n, nt = p.nt, m = p.m, N = p.N, wcap = p.wcap, LD = wcap + 1;
    const int TP = sdual_tp(NT, MC, N);
    double* L = smem + (size_t)wv * p.lds_per_wave;
    double* sA = L;
    double* sB = sA + TP;
    double* Sinv = sB + TP;                    // [wcap][LD]
    double* gbuf = Sinv + wcap * LD + (wcap * LD & 1);
    double* cbuf = gbuf + NT;
    double* ubuf = cbuf + wcap;
    const bool is_state_lane = lane < NT, is_input_lane = lane >= NT && lane < SP;
    const int rrow = lane < SP ? lane : SP - 1;   // record row of this lane (idle lanes shadow the last row; their results are dropped)
    const bool kvar = p.rec_kstride != 0;
    const bool has_box = p.xmin != nullptr, has_eq = p.eqt != nullptr, with_c = p.pc != nullptr;

    double r1[NT], sb[MC], r2[NT], r3[MC];
    auto load_bw = [&](const double* recg, int k) {
        const double* rk = recg + (size_t)k * p.rec_kstride + (size_t)rrow * RL;
#pragma unroll
        for (int j = 0; j < NT; j += 2) { const double2 t = *reinterpret_cast<const double2*>(rk + j); r1[j] = t.x; r1[j + 1] = t.y; }
#pragma unroll
        for (int j = 0; j < MC; j += 2) { const double2 t = *reinterpret_cast<const double2*>(rk + NT + j); sb[j] = t.x; sb[j + 1] = t.y; }
    };
    auto load_fw = [&](const double* recg, int k) {
        const double* rk = recg + (size_t)k * p.rec_kstride + (size_t)rrow * RL + NT + MC;
#pragma unroll
        for (int j = 0; j < NT; j += 2) { const double2 t = *reinterpret_cast<const double2*>(rk + j); r2[j] = t.x; r2[j + 1] = t.y; }
#pragma unroll
        for (int j = 0; j < MC; j += 2) { const double2 t = *reinterpret_cast<const double2*>(rk + NT + j); r3[j] = t.x; r3[j + 1] = t.y; }
    };
    if (!kvar && p.rec_stride == 0) { load_bw(p.rec, 0); load_fw(p.rec, 0); }

    const int wpb = (int)(blockDim.x >> 6);
    const int nwaves = gridDim.x * wpb;
    for (int inst = blockIdx.x * wpb + wv; inst < p.batch; inst += nwaves) {
        if (p.only_ovf) {
            if (p.ovf[inst] == 0) continue;
        } else if ((p.filter == 1 && p.status[inst] == 0) || (p.filter == 2 && p.status[inst] == 0 && p.flag[inst] == 0)) {
            if (p.ovf && lane == 0) p.ovf[inst] = 0;
            continue;
        }
        const double* recg = p.rec + (size_t)inst * p.rec_stride;
        const double* urg = p.uref + (size_t)inst * p.uref_stride;
        const double* xbr = p.xbref ? p.xbref + (size_t)inst * p.xbref_stride : nullptr;
        const double* xrg = p.xref ? p.xref + (size_t)inst * p.xref_stride : nullptr;
        const double* eqg = has_eq ? p.eqt + (size_t)inst * p.eqt_stride : nullptr;
        const double* pcg = with_c ? p.pc + (size_t)inst * p.pc_stride : nullptr;
        const double* ctg = with_c ? p.ct + (size_t)inst * p.pc_stride : nullptr;
        const double* bsg = p.base ? p.base + (size_t)inst * p.base_stride : nullptr;
        if (!kvar && p.rec_stride != 0) { load_bw(recg, 0); load_fw(recg, 0); }
        double* s = sA;    // row values of the current iterate (= the trajectory)
        double* w = sB;    // sources of a sweep, then its result
        double e0l = 0.0;  // e_0 (state lanes)
        if (lane < n) e0l = p.x0 ? p.x0[(size_t)inst * n + lane] - (xrg ? xrg[lane] : 0.0) : 0.0;

        // bounds of coordinate t (row: the coordinate carries bounds at all)
        auto bounds_of = [&](int t, double& lo, double& hi, bool& row, bool& iseq) {
            const int k = t / SP, j = t - k * SP;
            lo = -1e300; hi = 1e300; row = false; iseq = false;
            if (j >= NT) {
                const int a = j - NT;
                if (a < m && k < N) { const double ur = urg[k * m + a]; lo = p.umin[a] - ur; hi = p.umax[a] - ur; row = true; }
            } else if (j < n && k >= 1 && k <= N) {
                if (has_eq && k == N) { lo = hi = eqg[j]; row = true; iseq = true; }
                else if (has_box) { const double xr = xbr ? xbr[k * n + j] : 0.0; lo = p.xmin[j] - xr; hi = p.xmax[j] - xr; row = true; }
            }
        };
        auto width_of = [&](double lo, double hi) { return (lo > -1e299 && hi < 1e299 && hi > lo) ? hi - lo : 1.0; };

        // ---- one affine sweep over w (sources in, trajectory out).  kb: highest stage that carries a source; kend: forward stages 0..kend-1
        auto sweep = [&](int kb, int kend, bool full_problem) {
            // backward
            int kstart = N - 1;
            double gl = 0.0;
            if (kb >= N || full_problem) { if (is_state_lane) gl = w[N * SP + lane]; }
            else kstart = kb;
            if (full_problem && with_c && is_state_lane) gl += pcg[(size_t)(N - 1) * NT + lane];
            if (is_state_lane) gbuf[lane] = gl;
            sd_fence();
            for (int k = kstart; k >= 0; --k) {
                if (kvar) load_bw(recg, k);
                double g[NT], su[MC];
#pragma unroll
                for (int j = 0; j < NT; ++j) g[j] = gbuf[j];
#pragma unroll
                for (int j = 0; j < MC; ++j) su[j] = w[k * SP + NT + j];
                double acc0 = is_state_lane ? w[k * SP + lane] : 0.0, acc1 = 0.0;
#pragma unroll
                for (int j = 0; j < NT; j += 2) { acc0 = fma(r1[j], g[j], acc0); acc1 = fma(r1[j + 1], g[j + 1], acc1); }
#pragma unroll
                for (int j = 0; j < MC; j += 2) { acc0 = fma(sb[j], su[j], acc0); acc1 = fma(sb[j + 1], su[j + 1], acc1); }
                double out = acc0 + acc1;
                if (full_problem && with_c && k > 0 && is_state_lane) out += pcg[(size_t)(k - 1) * NT + lane];
                sd_fence();
                if (is_state_lane) gbuf[lane] = out;
                else if (is_input_lane) w[k * SP + lane] = out;   // kff_k over the consumed input sources
                sd_fence();
            }
            // forward
            if (is_state_lane) w[lane] = full_problem ? (lane < n ? e0l : 0.0) : 0.0;
            sd_fence();
            for (int k = 0; k < kend; ++k) {
                if (kvar) load_fw(recg, k);
                double xk[NT], kf[MC];
#pragma unroll
                for (int j = 0; j < NT; ++j) xk[j] = w[k * SP + j];
#pragma unroll
                for (int j = 0; j < MC; ++j) kf[j] = w[k * SP + NT + j];
                double acc0 = (full_problem && with_c && is_state_lane) ? ctg[(size_t)k * NT + lane] : 0.0, acc1 = 0.0;
#pragma unroll
                for (int j = 0; j < NT; j += 2) { acc0 = fma(r2[j], xk[j], acc0); acc1 = fma(r2[j + 1], xk[j + 1], acc1); }
#pragma unroll
                for (int j = 0; j < MC; j += 2) { acc0 = fma(r3[j], kf[j], acc0); acc1 = fma(r3[j + 1], kf[j + 1], acc1); }
                const double out = acc0 + acc1;
                sd_fence();
                if (is_state_lane) w[(k + 1) * SP + lane] = out;
                else if (is_input_lane) w[k * SP + lane] = out;
                sd_fence();
            }
        };
        auto zero_w = [&]() {
            for (int t = lane; t < TP; t += 64) w[t] = 0.0;
        };

        // ---- working set: position i lives on lane i
        int nW = 0, kmaxW = -1;
        int Wrow = 0, Wside = 0;
        double lam = 0.0, cpos = 0.0, rpos = 0.0;
        uint64_t inW = 0ull;   // bit e: coordinate e * 64 + lane is in the working set
        auto stage_of = [&](int t) { const int k = t / SP; return k < N ? k : N; };
        auto bound_of_pos = [&]() {   // bound the row of this lane's position sits on
            double lo, hi; bool row, iseq;
            bounds_of(Wrow, lo, hi, row, iseq);
            return Wside >= 0 ? hi : lo;
        };
        auto sinv_mul = [&](const double* vec) {   // (Sinv vec)_lane for lane < nW
            double a0 = 0.0, a1 = 0.0;
            const double* rowp = Sinv + (size_t)(lane < nW ? lane : 0) * LD;
            int j = 0;
            for (; j + 1 < nW; j += 2) { a0 = fma(rowp[j], vec[j], a0); a1 = fma(rowp[j + 1], vec[j + 1], a1); }
            if (j < nW) a0 = fma(rowp[j], vec[j], a0);
            return lane < nW ? a0 + a1 : 0.0;
        };
        // append row t (side sd); c in cpos (lanes < nW), u = Sinv c in rpos, gpp = Ghat_tt.  False: dependent on the rows already there.
        auto border = [&](int t, int sd, double gpp) {
            const double sc = gpp - sd_wave_sum(lane < nW ? cpos * rpos : 0.0);
            if (!(sc > 1e-12 * gpp)) return false;
            const double isc = 1.0 / sc;
            if (lane < nW) ubuf[lane] = rpos;
            sd_fence();
            if (lane < nW) {
                double* rowp = Sinv + (size_t)lane * LD;
                const double f = rpos * isc;
                for (int j = 0; j < nW; ++j) rowp[j] = fma(f, ubuf[j], rowp[j]);
                rowp[nW] = -f;
            } else if (lane == nW) {
                double* rowp = Sinv + (size_t)nW * LD;
                for (int j = 0; j < nW; ++j) rowp[j] = -ubuf[j] * isc;
                rowp[nW] = isc;
            }
            if (lane == nW) { Wrow = t; Wside = sd; lam = 0.0; }
            if (lane == (t & 63)) inW |= 1ull << (t >> 6);
            const int ks = stage_of(t);
            kmaxW = ks > kmaxW ? ks : kmaxW;
            ++nW;
            sd_fence();
            return true;
        };
        auto remove_pos = [&](int pos) {
            const int q = nW - 1;
            const int trow = __shfl(Wrow, pos);
            const double piv = Sinv[(size_t)pos * LD + pos];
            const double ipiv = 1.0 / piv;
            if (lane < nW && lane != pos) {
                double* rowp = Sinv + (size_t)lane * LD;
                const double f = rowp[pos] * ipiv;
                const double* prow = Sinv + (size_t)pos * LD;
                for (int j = 0; j < nW; ++j)
                    if (j != pos) rowp[j] = fma(-f, prow[j], rowp[j]);
            }
            sd_fence();
            if (pos != q) {   // the last position moves into the hole
                if (lane < q && lane != pos) Sinv[(size_t)lane * LD + pos] = Sinv[(size_t)lane * LD + q];
                sd_fence();
                if (lane == q) {
                    const double* qrow = Sinv + (size_t)q * LD;
                    double* prow = Sinv + (size_t)pos * LD;
                    for (int j = 0; j < q; ++j)
                        if (j != pos) prow[j] = qrow[j];
                    prow[pos] = qrow[q];
                }
                const int wr = __shfl(Wrow, q), ws = __shfl(Wside, q);
                const double wl = __shfl(lam, q), wc = __shfl(cpos, q);
                if (lane == pos) { Wrow = wr; Wside = ws; lam = wl; cpos = wc; }
            }
            if (lane == (trow & 63)) inW &= ~(1ull << (trow >> 6));
            --nW;
            kmaxW = sd_wave_max_i(lane < nW ? stage_of(Wrow) : -1);
            sd_fence();
        };
        // the trajectory for the current multipliers, from scratch: w = base + lam / 2 on the working-set rows, full sweep, s <-> w
        auto full_solve = [&]() {
            if (bsg) { for (int t = lane; t < TP; t += 64) w[t] = bsg[t]; }
            else zero_w();
            sd_fence();
            if (lane < nW) w[Wrow] += 0.5 * lam;
            sd_fence();
            sweep(N, N, true);
            double* t_ = s; s = w; w = t_;
        };
        // response of the unconstrained problem to a unit multiplier on coordinate t: w = Ghat[:, t] for stages < kend
        auto response = [&](int t, int kend) {
            zero_w();
            sd_fence();
            if (lane == 0) w[t] = -0.5;
            sd_fence();
            sweep(stage_of(t), kend, false);
        };

        int it = 0, status = 1;
        bool bad = false, overflow = false;
        // ---- stage 1 of the reference is x0 itself: outside the state box -> infeasible
        bool x0_out = false;
        if (has_box && lane < n) {
            const double xr = xbr ? xbr[lane] : 0.0;
            x0_out = e0l < p.xmin[lane] - xr || e0l > p.xmax[lane] - xr;
        }
        x0_out = __any(x0_out);
        full_solve();   // s = the unconstrained solution
        if (x0_out) status = 3;
        else {
            // ---- start: terminal-equality rows, then the guess's inputs on a bound
            auto try_add_start = [&](int t, int sd, int kend) {
                if (nW >= wcap) { overflow = true; return; }
                response(t, kend);
                if (lane < nW) cpos = w[Wrow];
                const double gpp = w[t];
                sd_fence();
                if (lane < nW) cbuf[lane] = cpos;
                sd_fence();
                rpos = sinv_mul(cbuf);
                (void)border(t, sd, gpp);
            };
            if (has_eq)
                for (int i = 0; i < n; ++i) try_add_start(N * SP + i, 0, N);
            if (p.uguess) {
                // highest stage of the guess's working set first (the forward passes of its responses stop there)
                int kg = -1;
                for (int t = lane; t < N * m; t += 64) {
                    const int k = t / m, a = t - k * m;
                    const double ur = urg[t], lo = p.umin[a] - ur, hi = p.umax[a] - ur;
                    double v = p.uguess[(size_t)inst * N * m + t] - ur;
                    v = v != v ? 0.0 : v;
                    if (hi > lo && (v >= hi || v <= lo)) kg = k > kg ? k : kg;
                }
                kg = sd_wave_max_i(kg);
                if (has_eq) kg = N;
                const int kend = kg + 1 < N ? kg + 1 : N;
                for (int t0 = 0; t0 < N * m && !overflow; t0 += 64) {   // trajectory order, 64 inputs per ballot
                    const int t = t0 + lane;
                    int sdl = 0;
                    if (t < N * m) {
                        const int k = t / m, a = t - k * m;
                        const double ur = urg[t], lo = p.umin[a] - ur, hi = p.umax[a] - ur;
                        double v = p.uguess[(size_t)inst * N * m + t] - ur;
                        v = v != v ? 0.0 : v;
                        if (hi > lo) sdl = v >= hi ? 1 : (v <= lo ? -1 : 0);
                    }
                    unsigned long long any = __ballot(sdl != 0);
                    const unsigned long long ups = __ballot(sdl > 0);
                    while (any && !overflow) {
                        const int b = __ffsll(any) - 1;
                        any &= any - 1;
                        const int tt = t0 + b, k = tt / m, a = tt - k * m;
                        try_add_start(k * SP + NT + a, ((ups >> b) & 1ull) ? 1 : -1, kend);
                    }
                }
            }
            if (nW > 0 && !overflow) {
                // multipliers of the start from s0; rows whose multiplier has the wrong sign leave, worst first
                while (nW > 0) {
                    const double b = lane < nW ? bound_of_pos() : 0.0;
                    if (lane < nW) cbuf[lane] = s[Wrow] - b;
                    sd_fence();
                    const double lw = sinv_mul(cbuf);
                    double viol = -__builtin_inf();
                    if (lane < nW && Wside != 0) viol = Wside > 0 ? -lw : lw;
                    const double lmax = fabs(lane < nW ? lw : 0.0);
                    double lm = lmax;
                    for (int o = 32; o > 0; o >>= 1) lm = fmax(lm, __shfl_xor(lm, o));
                    int vi = lane < nW ? Wrow : 0x7fffffff;
                    double vv = viol;
                    sd_argmax(vv, vi);
                    if (!(vv > 1e-12 * fmax(1.0, lm))) { lam = lane < nW ? lw : 0.0; break; }
                    const int pos = __ffsll((unsigned long long)__ballot(lane < nW && Wrow == vi)) - 1;
                    sd_fence();
                    remove_pos(pos);
                    ++it;
                }
                full_solve();
            }
            // ---- main loop
            int refined = 0;
            while (it < p.max_iter && !overflow) {
                ++it;
                // most violated row outside the working set (violation in units of the row's width; ties -> smallest coordinate)
                double vbest = -__builtin_inf();
                int tbest = 0x7fffffff;
                double chk = 0.0;
                for (int e = 0, t = lane; t < TP; ++e, t += 64) {
                    const double sv = s[t];
                    chk += sv - sv;
                    if ((inW >> e) & 1ull) continue;
                    double lo, hi; bool row, iseq;
                    bounds_of(t, lo, hi, row, iseq);
                    if (!row) continue;
                    const double vio = fmax(sv - hi, lo - sv) / width_of(lo, hi);
                    if (vio > vbest) { vbest = vio; tbest = t; }
                }
                if (__any(chk != 0.0)) { bad = true; break; }
                sd_argmax(vbest, tbest);
                if (!(vbest > p.tol)) {
                    // confirmation from scratch: working-set rows on their bounds, nothing else violated
                    full_solve();
                    double res = 0.0, resn = 0.0;
                    if (lane < nW) {
                        double lo, hi; bool row, iseq;
                        bounds_of(Wrow, lo, hi, row, iseq);
                        res = s[Wrow] - (Wside >= 0 ? hi : lo);
                        resn = fabs(res) / width_of(lo, hi);
                    }
                    double vmax = -__builtin_inf(), chk2 = 0.0;
                    for (int e = 0, t = lane; t < TP; ++e, t += 64) {
                        const double sv = s[t];
                        chk2 += sv - sv;
                        if ((inW >> e) & 1ull) continue;
                        double lo, hi; bool row, iseq;
                        bounds_of(t, lo, hi, row, iseq);
                        if (!row) continue;
                        vmax = fmax(vmax, fmax(sv - hi, lo - sv) / width_of(lo, hi));
                    }
                    if (__any(chk2 != 0.0)) { bad = true; break; }
                    for (int o = 32; o > 0; o >>= 1) { vmax = fmax(vmax, __shfl_xor(vmax, o)); resn = fmax(resn, __shfl_xor(resn, o)); }
                    if (resn <= 1e-8 && vmax <= 1e-8) { status = 0; break; }
                    if (refined >= 3) break;
                    ++refined;
                    if (nW > 0) {
                        if (lane < nW) cbuf[lane] = res;
                        sd_fence();
                        const double dl = sinv_mul(cbuf);
                        if (lane < nW) lam += dl;
                        full_solve();
                    }
                    continue;
                }
                const int pr = tbest;
                double plo, phi; bool prow_, peq;
                bounds_of(pr, plo, phi, prow_, peq);
                const int sd = s[pr] > phi ? 1 : -1;
                const double bp = sd > 0 ? phi : plo;
                if (nW >= wcap) { overflow = true; break; }
                double lam_p = 0.0;
                {
                    const int km = stage_of(pr) > kmaxW ? stage_of(pr) : kmaxW;
                    response(pr, km + 1 < N ? km + 1 : N);
                }
                if (lane < nW) cpos = w[Wrow];
                const double gpp = w[pr];
                sd_fence();
                bool infeasible = false;
                while (true) {
                    if (lane < nW) cbuf[lane] = cpos;
                    sd_fence();
                    rpos = sinv_mul(cbuf);
                    // direction in row space: Ghat (e_p - sum_w r_w e_w)
                    zero_w();
                    sd_fence();
                    if (lane == 0) w[pr] = -0.5;
                    if (lane < nW) w[Wrow] = 0.5 * rpos;
                    sd_fence();
                    {
                        const int kb = stage_of(pr) > kmaxW ? stage_of(pr) : kmaxW;
                        sweep(kb, N, false);
                    }
                    const double dp = w[pr];
                    const bool dependent = !(dp > 1e-12 * gpp);
                    const double tau2 = dependent ? __builtin_inf() : fabs(s[pr] - bp) / dp;
                    double t1 = __builtin_inf();
                    int bi = 0x7fffffff;
                    if (lane < nW && Wside != 0) {
                        const double ri = rpos * sd;
                        if (Wside > 0 && ri > 0.0) { t1 = fmax(lam, 0.0) / ri; bi = Wrow; }
                        else if (Wside < 0 && ri < 0.0) { t1 = fmin(lam, 0.0) / ri; bi = Wrow; }
                    }
                    sd_argmin(t1, bi);
                    const bool have_blk = bi != 0x7fffffff && t1 < __builtin_inf();
                    if (dependent && !have_blk) { infeasible = true; break; }
                    const double tau = fmin(t1, tau2);
                    if (!dependent) {
                        const double f = -(double)sd * tau;
                        for (int t = lane; t < TP; t += 64) s[t] = fma(f, w[t], s[t]);
                    }
                    if (lane < nW) lam -= (double)sd * tau * rpos;
                    lam_p += (double)sd * tau;
                    sd_fence();
                    if (have_blk && t1 < tau2) {
                        const int pos = __ffsll((unsigned long long)__ballot(lane < nW && Wrow == bi)) - 1;
                        remove_pos(pos);
                        ++it;
                        if (it >= p.max_iter) break;
                        continue;
                    }
                    if (border(pr, sd, gpp)) { if (lane == nW - 1) lam = lam_p; }
                    if (lane == 0) s[pr] = bp;
                    sd_fence();
                    break;
                }
                if (infeasible) { status = 3; break; }
            }
        }
        if (status == 3 && !p.rows_state) status = 1;   // an input box alone is never infeasible: a numerical verdict, not a certificate
        // ---- outputs
        const int st_out = bad ? 2 : status;
        if (p.ovf && lane == 0) p.ovf[inst] = (overflow && !bad) ? 1 : 0;
        for (int t = lane; t < N * m; t += 64) {
            const int k = t / m, a = t - k * m;
            const double ur = urg[t];
            const double v = s[k * SP + NT + a];
            double uu = (bad || st_out == 3) ? v + ur : fmin(fmax(v + ur, p.umin[a]), p.umax[a]);
            if (!p.v_only) p.u[(size_t)inst * N * m + t] = uu;
            p.eu[(size_t)inst * N * m + t] = uu - ur;
        }
        if (!p.v_only) {
            for (int t = lane; t < (N + 1) * n; t += 64) {
                const int k = t / n, i = t - k * n;
                const double e = k == 0 ? (p.x0 ? p.x0[(size_t)inst * n + i] - (xrg ? xrg[i] : 0.0) : 0.0) : s[k * SP + i];
                p.ex[(size_t)inst * (N + 1) * n + t] = e;
                p.x[(size_t)inst * (N + 1) * n + t] = (k == 0 && p.x0) ? p.x0[(size_t)inst * n + i] : e + (xrg ? xrg[t] : 0.0);
            }
        }
        if (lane == 0) {
            p.status[inst] = st_out;
            p.piters[inst] = it;
            if (p.filter == 2 && st_out == 0) p.flag[inst] = 0;
        }
        sd_fence();
    }
}

}  // namespace almpc
