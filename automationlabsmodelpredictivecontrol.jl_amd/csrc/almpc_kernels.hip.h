// almpc_kernels.hip.h -- gfx950 (CDNA4, MI355X) kernels of the per-step MPC solve.
//
// Path (reference -> here), reference paths relative to /root/reference:
//   update_initialization! + calculate!   src/main/computation_mpc.jl:17-55
//   OSQP ADMM iteration (third-party libosqp reached via JuMP.optimize!, computation_mpc.jl:41)
// for the condensed QP of SURVEY.md section 8a, in Jacobi-scaled coordinates v = D w:
//   min 1/2 w'H'w + f'w,  lo' <= w <= hi',  H' = DHD (shared by the batch), f' = D(F e0 + fS).
//
// Kernels
//   k_admm<NRB,KS>  gradient f' = F'e0 (MFMA), box ADMM loop with the shared KKT inverse
//                   (H'+(sigma+rho)I)^-1 register-resident as FP64 MFMA A-fragments, one 16-instance
//                   tile per workgroup; v0 = -H'^-1 f' for the polish with the same machinery.
//   k_polish        exact active-set finish, one wave per instance, (G_WW)^-1 in LDS (global past 32 rows).
//   k_rollout       u, e_u, x, e_x from w (recursive e+ = A e + B v).
//
// MFMA used: v_mfma_f64_16x16x4_f64.  Lane l holds A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
// C/D: col = l&15, row = (l>>4) + 4*reg  (cdna_hip_programming.md section 3, f64 map).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <utility>
#include <type_traits>

namespace almpc {

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-function, process-wide attribute (per device): keep the largest value ever
// asked for per (device, kernel) and only ever raise it, so that handles of different shapes cannot lower each other's limit.
inline hipError_t ensure_dyn_lds(const void* fn, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;  // within the default limit
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> cur;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t& c = cur[std::make_pair(dev, fn)];
    if (c >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) c = bytes;
    return e;
}


// Diagnostic build only (-DALMPC_STAMPS): lane 0 of every wave records the shader clock at phase boundaries into a
// debug buffer of its own (never read by kernel code).  The shipped library has no stamps.
#ifdef ALMPC_STAMPS
__device__ long long* g_stamps = nullptr;  // [waves][16]
#define ALMPC_STAMP(WAVE_ID, SLOT)                                                                     \
    do {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        if (g_stamps && (threadIdx.x & 63) == 0) g_stamps[(size_t)(WAVE_ID) * 16 + (SLOT)] = __builtin_readcyclecounter(); \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
#define ALMPC_ACC_DECL long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long acc_prev = 0; int acc_n = 0;
#define ALMPC_ACC_START do { __builtin_amdgcn_sched_barrier(0); acc_prev = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define ALMPC_ACC(SLOT) do { __builtin_amdgcn_sched_barrier(0); const long long n_ = __builtin_readcyclecounter(); acc_t[SLOT] += n_ - acc_prev; acc_prev = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define ALMPC_ACC_FLUSH(WAVE_ID) do { if (g_stamps && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 8; ++i_) g_stamps[(size_t)(4096 + (WAVE_ID)) * 16 + i_] = acc_t[i_]; g_stamps[(size_t)(4096 + (WAVE_ID)) * 16 + 8] = acc_n; } } while (0)
#else
#define ALMPC_STAMP(WAVE_ID, SLOT) do { } while (0)
#define ALMPC_ACC_DECL
#define ALMPC_ACC_START do { } while (0)
#define ALMPC_ACC(SLOT) do { } while (0)
#define ALMPC_ACC_FLUSH(WAVE_ID) do { } while (0)
#endif

// Packed lower triangle of a symmetric nz x nz matrix, column by column: element (i, c), i >= c, at packed_tri_off(nz, c) + i - c.
// A column is contiguous in i, so the inverse kernels write it coalesced (k_design_inverse_c32, packed_out) and a reader that wants
// element (r, c) of the full matrix takes (max, min).  packed_tri_doubles: per-instance stride, rounded up to whole 16-byte pairs.
// Columns are padded so that packed_tri_off(c) - c is even: element (r, c) with r even then sits on a 16-byte boundary, and so does
// (c, r') read as column c's entries r', r' + 1 for even r' -- every 2 x 2 block a lane of k_admm_inst<true> gathers is two
// aligned 16-byte LDS reads, whichever side of the diagonal it lies on.
__host__ __device__ inline int packed_tri_off(int nz, int c) {
    // column c holds nz - c entries; an odd-numbered column of an even-sized matrix (and vice versa) is followed by one pad double
    // exactly when that keeps off(c + 1) - (c + 1) even.  Closed form: off(c) = sum_{k<c} (nz - k + pad_k), pad_k = (nz - k + 1) & 1
    // with off(0) = 0 (even): off(k+1) - (k+1) = off(k) - k + (nz - k + pad_k) - 1 stays even iff nz - k + pad_k is odd.
    const int full = c * nz - (c * (c - 1)) / 2;
    // pad_k = 1 when (nz - k) is even: k in [0, c) with k == nz (mod 2)
    const int pads = (nz & 1) ? (c / 2) : ((c + 1) / 2);
    return full + pads;
}
__host__ __device__ inline long packed_tri_doubles(int nz) { return ((long)packed_tri_off(nz, nz) + 1) & ~1L; }

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2v __attribute__((ext_vector_type(2)));

constexpr int TILE = 16;  // instances per workgroup tile = MFMA N dimension

__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// D(16 x 16) = sum_ks Afrag[ks] (16 x 4) * Bbuf[4ks..4ks+3][0..15]; Bbuf is an LDS image [k][16].
// Two accumulators so that consecutive MFMAs are independent.
template <int KS>
__device__ __forceinline__ d4 tile_matmul(const double (&a)[KS], const double* bbuf, int q, int col) {
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
        double b0 = bbuf[(4 * ks + q) * TILE + col];
        acc0 = mfma_f64(a[ks], b0, acc0);
        if (ks + 1 < KS) {
            double b1 = bbuf[(4 * (ks + 1) + q) * TILE + col];
            acc1 = mfma_f64(a[ks + 1], b1, acc1);
        }
    }
    return acc0 + acc1;
}

__device__ __forceinline__ double qmax(double v) {  // max over the 4 lanes that share an instance column
    v = fmax(v, __shfl_xor(v, 16));
    v = fmax(v, __shfl_xor(v, 32));
    return v;
}

struct AdmmParams {
    int nz, n, m, batch, nzs;  // nzs: row stride of per-instance vectors (= 16*NRB)
    const double* MinvFrag;    // [NRB][KS][64]   (H' + (sigma+rho) I)^-1
    const double* VFrag;       // [NRB][ksf][64]  -H'^-1 F'  (v0 = V e0 + v0S)
    const double* v0S;         // [1 or batch][nz] -H'^-1 fS
    long v0S_stride;
    const double* HFrag;       // [NRB][KS][64]   H'   (warm start only)
    const double* FFrag;       // [NRB][KSF][64]  F' = D F
    int ksf;
    const double* dvec;        // [nzs] scaling d (pad rows: 1)
    const double* rhovec;      // [nzs] ADMM penalty per row (scalar rho, or the stiffness profile rho/G_ii)
    const double* umin;        // [m]
    const double* umax;        // [m]
    const double* uref;        // [uref_stride*inst + row]
    long uref_stride;
    const double* xref;        // first column: [xref_stride*inst + k]
    long xref_stride;
    const double* fS;          // [fS_stride*inst + row], scaled
    long fS_stride;
    const double* x0;          // [batch][n]
    double* xs;                // ADMM state, scaled coordinates, [batch][nzs]
    double* zs;
    double* ys;
    double* v0;                // -H'^-1 f', [batch][nzs]
    int32_t* status;
    int32_t* iters;
    int32_t* piters;    // zeroed here so that the step needs no memset nodes
    int32_t* perm;      // [tiles*16] polish processing order inside each tile: large active-set guess first (see k_polish)
    double rho, sigma, alpha, eps_abs, eps_rel;
    int max_iter, check_every, warm;
    // keep_state = 0 (opts.reserved[0] & ALMPC_OPT_NO_WARM_STATE): the ADMM state is not kept for a warm start of the next step: x and y
    // are not written at all; of y the polish needs only the signs, which go out as one 32-bit word per (instance, wave):
    // yflags[inst * NRB + w], bit b: y < 0 on row 16 w + b, bit 16 + b: y > 0.
    int keep_state;
    uint32_t* yflags;
};

// One workgroup = NRB waves = one tile of 16 instances.  Wave w owns rows 16w..16w+15 of every
// instance vector in the MFMA C/D layout: lane (q = l>>4, col = l&15) holds rows 16w + q + 4i,
// i = 0..3, of instance col -- so the whole ADMM vector update is register-local and only the
// right-hand side travels through LDS (double-buffered, one barrier per iteration).
// `after_requests` runs right after the prologue's own global loads have been requested (the fused step kernel asks for G there:
// loads return in order, so a stream requested first would hold up e0 and the fragments of the first iteration)
template <int NRB, int KS, typename AfterRequests>
__device__ __forceinline__ void admm_body(const AdmmParams& p, double* smem, AfterRequests after_requests) {
    constexpr int RP = 16 * NRB;  // padded rows
    static_assert(4 * KS <= RP, "K padding must fit the row padding");
    double* rhs0 = smem;                    // [RP][16]
    double* rhs1 = smem + RP * TILE;        // [RP][16]
    double* red = smem + 2 * RP * TILE;     // [NRB][8][16]
    double* e0s = red + NRB * 8 * TILE;     // [4*ksf][16]

    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = lane & 15, q = lane >> 4;
    const int inst = blockIdx.x * TILE + col;
    const bool valid = inst < p.batch;
    const int instc = valid ? inst : p.batch - 1;  // clamp: pad columns recompute the last instance, never stored
    ALMPC_STAMP(blockIdx.x * NRB + wv, 0);

    // ---- every other global load of the prologue is requested up front as well (one exposed latency, not five):
    // F' fragments, the row constants, and x0 / x_ref for e0
    constexpr int KSF_MAX = 16;  // n <= 64
    double af[KSF_MAX];
#pragma unroll
    for (int ks = 0; ks < KSF_MAX; ++ks)
        af[ks] = (ks < p.ksf) ? p.FFrag[((size_t)(wv * p.ksf + ks)) * 64 + lane] : 0.0;
    int row[4];
    double dv[4], dinv[4], lo[4], hi[4], fs[4], rho[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        row[i] = wv * 16 + q + 4 * i;
        const bool in = row[i] < p.nz;
        const int r = in ? row[i] : 0;
        dv[i] = in ? p.dvec[r] : 1.0;
        rho[i] = p.rhovec[r];  // per-row penalty, requested with the rest of the prologue's loads
        const double ur = p.uref[(size_t)instc * p.uref_stride + r];
        lo[i] = p.umin[r % p.m] - ur;   // scaled below
        hi[i] = p.umax[r % p.m] - ur;
        fs[i] = in ? p.fS[(size_t)instc * p.fS_stride + r] : 0.0;
    }
    // ---- e0 = x0 - x_ref[:,1] into LDS as B operand [k][16]
    const int kpf = 4 * p.ksf;
    for (int t = threadIdx.x; t < kpf * TILE; t += blockDim.x) {
        int k = t / TILE, c = t % TILE;
        int ii = blockIdx.x * TILE + c;
        if (ii >= p.batch) ii = p.batch - 1;
        double v = 0.0;
        if (k < p.n) v = p.x0[(size_t)ii * p.n + k] - p.xref[(size_t)ii * p.xref_stride + k];
        e0s[t] = v;
    }
    // ---- shared KKT inverse -> registers (A fragments), coalesced 512 B per wave-instruction.  Requested LAST: loads return
    // in order, and these 30 fragments (115 KB per workgroup) are not needed before the first iteration, while everything
    // above is needed by the prologue, which now runs under this stream instead of behind it.
    double a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = p.MinvFrag[((size_t)(wv * KS + ks)) * 64 + lane];
    after_requests();
    ALMPC_STAMP(blockIdx.x * NRB + wv, 5);
    __syncthreads();
    ALMPC_STAMP(blockIdx.x * NRB + wv, 6);

    // ---- per-row constants and f' = F' e0 + fS
    {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KSF_MAX; ++ks)
            if (ks < p.ksf) acc = mfma_f64(af[ks], e0s[(4 * ks + q) * TILE + col], acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = row[i] < p.nz;
            dinv[i] = 1.0 / dv[i];
            lo[i] = in ? lo[i] * dinv[i] : 0.0;
            hi[i] = in ? hi[i] * dinv[i] : 0.0;
            fs[i] = in ? acc[i] + fs[i] : 0.0;
        }
    }

    ALMPC_STAMP(blockIdx.x * NRB + wv, 7);
    // ---- initial iterate
    // y is carried in scaled form yt = y / rho_i (the update needs no 1/rho then); rho_i is per row
    double x[4], z[4], yt[4], px[4], rown[4];
    const double sigma = p.sigma, alpha = p.alpha;
    if (p.warm) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // all twelve loads first (one exposed latency), then the arithmetic and the LDS writes
            const size_t o = (size_t)instc * p.nzs + row[i];
            x[i] = p.xs[o];
            yt[i] = p.ys[o];
            z[i] = p.zs[o];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            yt[i] = yt[i] / rho[i];
            z[i] = fmin(fmax(z[i], lo[i]), hi[i]);
            rhs0[row[i] * TILE + col] = x[i];
        }
        __syncthreads();
        // px = H' x needs one true product: stream the H' fragments once
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        // (eight fragments per L2 round trip: one at a time the compiler waited for each load, ~190 cycles x KS)
#pragma unroll
        for (int ks0 = 0; ks0 < KS; ks0 += 8) {
            double ah[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (ks0 + u < KS) ah[u] = p.HFrag[((size_t)(wv * KS + ks0 + u)) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (ks0 + u < KS) acc = mfma_f64(ah[u], rhs0[(4 * (ks0 + u) + q) * TILE + col], acc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) px[i] = acc[i];
        __syncthreads();
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = z[i] = yt[i] = px[i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rown[i] = sigma * x[i] - fs[i] + rho[i] * (z[i] - yt[i]);
        rhs0[row[i] * TILE + col] = rown[i];
    }
    // |f/d|_inf per instance (constant part of the dual tolerance)
    {
        double mf = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) mf = fmax(mf, fabs(fs[i] * dinv[i]));
        mf = qmax(mf);
        if (q == 0) red[(wv * 8 + 7) * TILE + col] = mf;
    }
    __syncthreads();
    double nf = 0.0;
    for (int w2 = 0; w2 < NRB; ++w2) nf = fmax(nf, red[(w2 * 8 + 7) * TILE + col]);
    __syncthreads();

    ALMPC_STAMP(blockIdx.x * NRB + wv, 1);
    bool active = true;
    int my_iters = p.max_iter, my_status = 1;
    double* cur = rhs0;
    double* nxt = rhs1;
    for (int it = 1; it <= p.max_iter; ++it) {
        const d4 xt4 = tile_matmul<KS>(a, cur, q, col);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double xt = xt4[i];
            if (active) {
                const double hxt = rown[i] - (sigma + rho[i]) * xt;  // H' xt, from the KKT identity
                px[i] = alpha * hxt + (1.0 - alpha) * px[i];
                x[i] = alpha * xt + (1.0 - alpha) * x[i];
                const double w = alpha * xt + (1.0 - alpha) * z[i] + yt[i];
                const double zn = fmin(fmax(w, lo[i]), hi[i]);
                yt[i] = w - zn;
                z[i] = zn;
                rown[i] = sigma * x[i] - fs[i] + rho[i] * (z[i] - yt[i]);
            }
            nxt[row[i] * TILE + col] = rown[i];
        }
        const bool check = (it % p.check_every == 0) || (it == p.max_iter);
        if (check) {
            double m_rp = 0, m_x = 0, m_z = 0, m_rd = 0, m_hx = 0, m_y = 0, m_bad = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                m_rp = fmax(m_rp, fabs(dv[i] * (x[i] - z[i])));
                m_x = fmax(m_x, fabs(dv[i] * x[i]));
                m_z = fmax(m_z, fabs(dv[i] * z[i]));
                const double yi = rho[i] * yt[i];
                m_rd = fmax(m_rd, fabs((px[i] + fs[i] + yi) * dinv[i]));
                m_hx = fmax(m_hx, fabs(px[i] * dinv[i]));
                m_y = fmax(m_y, fabs(yi * dinv[i]));
                const double s = x[i] + yi + px[i];
                if (!(fabs(s) <= 1.79e308)) m_bad = 1.0;
            }
            m_rp = qmax(m_rp); m_x = qmax(m_x); m_z = qmax(m_z); m_rd = qmax(m_rd);
            m_hx = qmax(m_hx); m_y = qmax(m_y); m_bad = qmax(m_bad);
            if (q == 0) {
                double* r = red + wv * 8 * TILE + col;
                r[0 * TILE] = m_rp; r[1 * TILE] = m_x; r[2 * TILE] = m_z; r[3 * TILE] = m_rd;
                r[4 * TILE] = m_hx; r[5 * TILE] = m_y; r[6 * TILE] = m_bad;
            }
        }
        __syncthreads();
        if (check) {
            double rp = 0, nx = 0, nzn = 0, rd = 0, nhx = 0, ny = 0, bad = 0;
            for (int w2 = 0; w2 < NRB; ++w2) {
                const double* r = red + w2 * 8 * TILE + col;
                rp = fmax(rp, r[0 * TILE]); nx = fmax(nx, r[1 * TILE]); nzn = fmax(nzn, r[2 * TILE]);
                rd = fmax(rd, r[3 * TILE]); nhx = fmax(nhx, r[4 * TILE]); ny = fmax(ny, r[5 * TILE]);
                bad = fmax(bad, r[6 * TILE]);
            }
            if (active) {
                const bool conv = (rp <= p.eps_abs + p.eps_rel * fmax(nx, nzn)) &&
                                  (rd <= p.eps_abs + p.eps_rel * fmax(fmax(nhx, ny), nf));
                if (bad > 0.0) { active = false; my_iters = it; my_status = 2; }
                else if (conv) { active = false; my_iters = it; my_status = 0; }
            }
            const bool any_active = __any(active);  // every wave sees all 16 columns -> same answer
            __syncthreads();                        // red is reused by the next check
            if (!any_active) { double* t = cur; cur = nxt; nxt = t; break; }
        }
        double* t = cur; cur = nxt; nxt = t;
    }

    ALMPC_STAMP(blockIdx.x * NRB + wv, 2);
    // ---- v0 = -H'^-1 f' for the polish.  f' = F' e0 + fS is affine in e0, so v0 = V e0 + v0S with V = -H'^-1 F' (design)
    // and v0S = -H'^-1 fS (set_reference): n columns instead of a second nz x nz tile product.  Loads issued before the stores.
#pragma unroll
    for (int ks = 0; ks < KSF_MAX; ++ks)
        af[ks] = (ks < p.ksf) ? p.VFrag[((size_t)(wv * p.ksf + ks)) * 64 + lane] : 0.0;
    double v0s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v0s[i] = (row[i] < p.nz) ? p.v0S[(size_t)instc * p.v0S_stride + row[i]] : 0.0;
    if (valid && wv == 0 && q == 0) {
        p.iters[inst] = my_iters;
        p.status[inst] = my_status;
        p.piters[inst] = 0;
    }
    // ---- polish order: the polish is bound by its slowest instances, so within every tile the instances are ranked by the
    // size of their active-set guess (a proxy for a long active-set chain): perm[tile*16 + rank] = instance (-1: pad column).
    // The polish hands out rank 0 of every tile first, then rank 1, ...  No global counters or atomics are involved.
    {
        int cntf = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) cntf += (row[i] < p.nz && yt[i] != 0.0) ? 1 : 0;
        cntf += __shfl_xor(cntf, 16);
        cntf += __shfl_xor(cntf, 32);
        if (q == 0) red[(wv * 8 + 0) * TILE + col] = (double)cntf;
    }
    __syncthreads();  // everyone is done reading cur/nxt; red holds the per-wave counts
    if (wv == 0 && q == 0) {  // lanes 0..15 of wave 0: one instance each
        int k0 = valid ? 0 : -1;  // pad columns sort last
        if (valid)
            for (int w2 = 0; w2 < NRB; ++w2) k0 += (int)red[(w2 * 8 + 0) * TILE + col];
        red[(0 * 8 + 1) * TILE + col] = (double)k0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int rank = 0;
        for (int c2 = 0; c2 < TILE; ++c2) {
            const int ko = (int)red[(0 * 8 + 1) * TILE + c2];
            rank += (ko > k0 || (ko == k0 && c2 < col)) ? 1 : 0;
        }
        p.perm[blockIdx.x * TILE + rank] = valid ? inst : -1;
    }
    d4 v04 = {v0s[0], v0s[1], v0s[2], v0s[3]};
#pragma unroll
    for (int ks = 0; ks < KSF_MAX; ++ks)
        if (ks < p.ksf) v04 = mfma_f64(af[ks], e0s[(4 * ks + q) * TILE + col], v04);
    ALMPC_STAMP(blockIdx.x * NRB + wv, 3);

    // ---- results (scaled coordinates) to HBM: transpose each [row][instance] register tile through LDS so that every
    // instance vector (nzs contiguous doubles) leaves as full 16-byte-per-lane coalesced stores
    constexpr int TS = RP + 2;  // staging stride per instance: even (16-byte pairs) and conflict-free for the scatter
    static_assert(TILE * TS <= 2 * RP * TILE, "staging tile must fit the two rhs buffers");
    double* stage = smem;
    auto flush = [&](const double (&v)[4], double* dst) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) stage[col * TS + row[i]] = v[i];
        __syncthreads();
        for (int e = 2 * threadIdx.x; e < TILE * RP; e += 2 * blockDim.x) {
            const int ci = e / RP, r = e % RP;
            const int ii = blockIdx.x * TILE + ci;
            if (ii < p.batch)
                *reinterpret_cast<d2v*>(dst + (size_t)ii * p.nzs + r) = *reinterpret_cast<const d2v*>(stage + ci * TS + r);
        }
    };
    const double v0r[4] = {v04[0], v04[1], v04[2], v04[3]};
    double y[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = rho[i] * yt[i];
    if (p.keep_state) {
        flush(x, p.xs);
        flush(y, p.ys);
    } else {
        uint32_t mk = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t b = (uint32_t)(q + 4 * i);
            mk |= (yt[i] < 0.0 ? 1u : 0u) << b;
            mk |= (yt[i] > 0.0 ? 1u : 0u) << (16u + b);
        }
        mk |= (uint32_t)__shfl_xor((int)mk, 16);
        mk |= (uint32_t)__shfl_xor((int)mk, 32);
        if (q == 0 && valid) p.yflags[(size_t)inst * NRB + wv] = mk;
    }
    flush(z, p.zs);
    flush(v0r, p.v0);
    ALMPC_STAMP(blockIdx.x * NRB + wv, 4);
}

template <int NRB, int KS>
__global__ __launch_bounds__(64 * NRB) void k_admm(AdmmParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    admm_body<NRB, KS>(p, smem, []() {});
}

// ------------------------------------------------------------------------------------------------
// polish: primal active-set finish with the shared inverse G = H'^-1 (oracle: polish_active_set)
// ------------------------------------------------------------------------------------------------
struct RolloutParams {
    int n, m, N, batch, nzs;
    const double* A;  // n*n column-major
    const double* B;  // n*m column-major
    long A_stride = 0, B_stride = 0, d_stride = 0;  // per-instance models: doubles between instances (k_rollout<1> only); 0 = shared
    const double* dvec;
    const double* w;  // scaled solution [batch][nzs]
    const double* x0;
    const double* xref;  // [xref_stride*inst + n*k + i]
    long xref_stride;
    const double* uref;
    long uref_stride;
    const double* umin;  // [m]: u is clamped to the box after un-scaling (d*(b/d) may be 1 ulp off b)
    const double* umax;
    double* x;   // [batch][N+1][n]
    double* ex;
    double* u;   // [batch][N][m]
    double* eu;
};

struct PolishParams {
    int nz, m, batch, nzs;
    const double* G;     // dense [nz][nzs], symmetric
    const double* dvec;  // [nzs]
    const double* umin;
    const double* umax;
    const double* uref;
    long uref_stride;
    const double* zs;    // ADMM z, y (scaled)
    const double* ys;
    const uint32_t* yflags;  // non-null: the signs of y as AdmmParams::yflags (ys is then not read), yflag_words words per instance
    int yflag_words;
    const double* v0;
    double* w;           // result (scaled), [batch][nzs]
    int32_t* status;     // in: ADMM status; out: final
    int32_t* piters;
    double* sglobal;     // [batch][64*64] scratch for working sets beyond 32 rows
    const int32_t* start_rows = nullptr;   // START builds (k_polish_sgl<1>), or null: [batch][65] count + rows (ascending) of a guessed working set of
                                           // 33..64 rows whose inverse k_guess_iterate_ws has left in sglobal (64 x 64, identity padded): installed at once
    const int32_t* perm; // [ntiles*16] per-tile processing order written by k_admm (hard instances first, -1 = pad)
    int ntiles;          // ADMM tiles (16 instances each)
    int lds_per_wave;    // doubles of LDS per wave (>= POLISH_LDS_MIN_PER_WAVE and >= the rollout trajectory buffer)
    // per-instance models (almpc_design_batched): strides in doubles of G, d, A, B per instance (0 = shared), and the
    // offset inside the wave's LDS slot of its private copy of d_i | [A_i B_i] (which replace the workgroup-shared ones)
    long G_stride, d_stride, A_stride, B_stride;
    int wave_const_off;
    int sg_off;          // k_polish_sgl: offset inside the wave's LDS slot of its 64 x 64 Sinv (working sets beyond 32 rows)
    int sg_shared_off;   // G-in-LDS builds: offset (doubles, from the shared constants) of ONE workgroup-shared second-tier slot of
                         // POLISH_SG_SHARED_CAP columns x 64 rows, or < 0: a wave whose working set outgrows 32 rows claims it (LDS
                         // word beside the queue counter) and otherwise falls back to the global scratch
    int g_off;           // k_polish_sgl: offset of the wave's copy of its instance's G_i (nz rows of nzs doubles)
    int max_iter;
    const int* dflag = nullptr;   // design flags (or null): an instance whose flag is set leaves with ALMPC_NON_FINITE (re-linearisation pipeline)
    int* unsolved = nullptr;  // host-visible counter (or null): += 1 for every instance that leaves the finish with status != 0 (lazy redo, almpc_api.hip)
    int* redo_gate = nullptr; // device word (or null): = step_serial when this step leaves an instance with status != 0 -- what a redo launch
    int step_serial = 0;      // that was enqueued behind the step without a host look tests before it does anything (SdualParams::gate)
    int direct = 0;      // k_step_inst_wave: workgroup (= wave) b finishes instance b itself (no perm lookup)
    int fuse_rollout;    // 1: this kernel also produces u, e_u, x, e_x (roll.*), no separate k_rollout launch; 2: u, e_u only;
                         // 3: as 1 with the blocked rollout (shared model: rollM)
    int roll_g, roll_cpl; // rollout lane decomposition: roll_g lanes per state row, roll_cpl columns of [A B] per lane
    // blocked rollout (fuse_rollout == 3): roll_s stages per block, one lane per (stage in block, state row); rollM
    // [ROLL_SMX + ROLL_NX][64] holds lane (j, i)'s row of [Gamma_s | Phi_s], zero padded: ROLL_SMX input columns (A^(j-t) B for
    // t <= j), then ROLL_NX state columns (A^(j+1))
    const double* rollM;
    int roll_s, roll_nb;
    RolloutParams roll;
};

__device__ __forceinline__ double readlane_d(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// 1 / v by v_rcp_f64 (2^-24) and two Newton steps: ~43 cycles on a dependent chain against ~90 for the IEEE division sequence
// (div_scale, rcp, Newton, div_fmas, div_fixup); the last bit may differ.  tools/microbench/dep_latency.hip.
__device__ __forceinline__ double fast_rcp_d(double v) {
    double r = __builtin_amdgcn_rcp(v);
    r = __builtin_fma(__builtin_fma(-v, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-v, r, 1.0), r, r);
    return r;
}

// One pivot of the symmetric Gauss-Jordan sweep of a matrix held one ROW per lane, this wave holding 16 of its columns (c0 .. c0 + 15)
// in registers (k_sdual_start, k_guess_iterate_ws: the waves of a workgroup share the columns).  The pivot row has been published
// through LDS by the pivot lanes and read by the caller in one batch: pj = this wave's 16 entries of it (wave-uniform reads), col = the
// lane's own entry, which is also its element of the pivot COLUMN, the matrix being symmetric, d = the pivot.  (Measured and dropped:
// taking the 16 entries out of `col` with v_readlane instead of reading them from LDS -- 41.6 -> 48.7 us for the 50-row sets of the
// SQP iteration: a readlane feeding a vector operand costs its ~38-cycle link 32 times per pivot.)
// Sweep convention: after the rows K, M_KK = -(G_KK)^-1, M_iK = G_iK (G_KK)^-1, the rest the Schur complement.  Row i != k:
// r_ij -= (col_i / d) pj_j; the pivot row: pj_j / d (= 0 - (-1/d) pj_j: the lane's old row is masked by `keep`); column k: col_i / d,
// and -1/d on the pivot (the pivot lane's factor).
typedef double gj16_row[16];
// (Measured and dropped, round 5: the rows as an ext_vector_type(16) with the pivot columns written at a wave-uniform DYNAMIC index.
// The compiler if-converts the guarded write into an indexed move on a COPY of the array followed by a select per register -- no
// fewer instructions -- and executes the indexed move also when the guard is false, with an index outside the array: a GPU memory
// fault on the first run.  A select per register on a scalar condition it is.)
// JJ: the register that holds column k, a COMPILE-TIME index (the callers unroll their pivot loop by 16: pivot k sits at position k & 15 of
// the wave that owns it); own: this wave holds column k (wave-uniform).  With the index known the pivot column is one guarded move --
// as a select per register on `jj == k - c0` it was 4 vector + 4 scalar instructions for each of the 16 registers, and per PAIR of
// pivots twice that: 130 of the 230 instructions of a step (2.3 k cycles per step measured; the sweep is bound by its instruction count).
template <int JJ>
__device__ __forceinline__ void gj16_pivot(gj16_row& r, const double (&pj)[16], const double col, const double d, const bool own, const int k,
                                           const int lane) {
    const double invd = fast_rcp_d(d);
    const bool piv = lane == k;
    const double f = piv ? -invd : col * invd;
    const double keep = piv ? 0.0 : 1.0;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) r[jj] = __builtin_fma(-f, pj[jj], keep * r[jj]);
    if (own) r[JJ] = f;   // column k: f on every lane -- the pivot lane's f IS -1/d
}

// Two pivots (k, k + 1) of the same sweep behind ONE publication and barrier: sweeping k and then k + 1 is the block sweep on the pair,
//   [f0 f1] = [col0 col1] D^-1,  D = [d11 d12; d12 d22] (entries k, k + 1 of the two published rows),  r_ij -= f0 pj0_j + f1 pj1_j,
// with D^-1 from the sequential formulas (i11 = 1 / d11, t = d12 i11, s22 = d22 - t d12, then 1 / s22: the second pivot as the one-by-one
// sweep forms it; the caller has computed them for its pivot tests).  The two pivot rows become D^-1 [row_k; row_k+1]: the same FMAs
// with (f0, f1) = -(row of D^-1) and the lane's old row masked; columns k and k + 1 of every row are (f0, f1).  JJ even: both columns
// sit on the same wave.
template <int JJ>
__device__ __forceinline__ void gj16_pivot2(gj16_row& r, const double (&pj0)[16], const double (&pj1)[16], const double col0, const double col1,
                                            const double i11, const double t, const double s22, const bool own, const int k, const int lane) {
    static_assert((JJ & 1) == 0 && JJ + 1 < 16, "a pair starts at an even position");
    const double e11 = fast_rcp_d(s22);
    const double e01 = -t * e11;
    const double e00 = __builtin_fma(-t, e01, i11);   // 1 / d11 + t^2 / s22
    const bool p0 = lane == k, p1 = lane == k + 1;
    double f0 = __builtin_fma(col0, e00, col1 * e01), f1 = __builtin_fma(col0, e01, col1 * e11);
    f0 = p0 ? -e00 : (p1 ? -e01 : f0);
    f1 = p0 ? -e01 : (p1 ? -e11 : f1);
    const double keep = (p0 || p1) ? 0.0 : 1.0;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) r[jj] = __builtin_fma(-f1, pj1[jj], __builtin_fma(-f0, pj0[jj], keep * r[jj]));
    if (own) { r[JJ] = f0; r[JJ + 1] = f1; }
}

// f(integral_constant<int, I>) for I = 0 .. N - 1, until it returns false
template <int I, int N, class F>
__device__ __forceinline__ void gj_static_for(F&& f) {
    if constexpr (I < N) {
        if (!f(std::integral_constant<int, I>{})) return;
        gj_static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---- wave-wide reductions on the DPP cross-lane path (no LDS crossbar): row_ror butterflies inside each 16-lane
// row, then row_bcast15 / row_bcast31 to fold the four rows; lane 63 holds the result, returned wave-uniform.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v) {  // lanes without a valid source (or masked rows) keep v
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
#define ALMPC_WAVE_REDUCE(NAME, OP)                                                     \
    __device__ __forceinline__ double NAME(double v) {                                  \
        v = OP(v, dpp_d<0x121, 0xF>(v)); /* row_ror:1 */                                \
        v = OP(v, dpp_d<0x122, 0xF>(v)); /* row_ror:2 */                                \
        v = OP(v, dpp_d<0x124, 0xF>(v)); /* row_ror:4 */                                \
        v = OP(v, dpp_d<0x128, 0xF>(v)); /* row_ror:8 */                                \
        { const double o = dpp_d<0x142, 0xA>(v); /* row_bcast15 -> rows 1,3 */          \
          const bool take = ((threadIdx.x >> 4) & 1) != 0; v = take ? OP(v, o) : v; }   \
        { const double o = dpp_d<0x143, 0xC>(v); /* row_bcast31 -> rows 2,3 */          \
          const bool take = ((threadIdx.x >> 5) & 1) != 0; v = take ? OP(v, o) : v; }   \
        return readlane_d(v, 63);                                                       \
    }
__device__ __forceinline__ double op_add(double a, double b) { return a + b; }
__device__ __forceinline__ double op_min(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ double op_max(double a, double b) { return fmax(a, b); }
ALMPC_WAVE_REDUCE(wave_sum, op_add)
ALMPC_WAVE_REDUCE(wave_min, op_min)
ALMPC_WAVE_REDUCE(wave_max, op_max)
#undef ALMPC_WAVE_REDUCE

// x(lane) + x(lane ^ 32) in every lane: two v_permlane32_swap (gfx950) instead of an LDS-path ds_bpermute
__device__ __forceinline__ double half_sum(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL load (vmcnt(0)),
// which would make a barrier in a compute phase wait for data that was deliberately requested early.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// generic -> global -> generic: tells the optimiser that a pointer it cannot trace back to the kernel arguments is a
// global one (global_load instead of flat_load, which would also tie up the LDS wait counter)
template <class T>
__device__ __forceinline__ T* GL(T* q) {
    return (T*)(__attribute__((address_space(1))) T*)q;
}

__device__ __forceinline__ void wave_fence_lds() {  // order this wave's LDS writes before its later LDS reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One wave per instance.  nz <= 128: lane l holds rows l and l+64 of the instance vectors ("row-distributed").
// "Position-distributed" quantities (row index, bound, side, multiplier lam, bordering vectors) live in registers,
// position i of the working set W on lane i.  Sinv = (G_WW)^-1 is stored column-major (S[c*LD + r]: a sweep over
// columns reads consecutive addresses across lanes) in one of two modes:
//   register mode (|W| <= 32): Sinv in registers, zero padded: lane (pos = lane & 31, hf = lane >> 5) holds
//                           Sinv[pos][16 hf .. 16 hf + 15]; both half-waves mirror the positions.  A product with Sinv
//                           is 16 FMAs on broadcast operands, a bordering or down-date is ONE FMA per register (the
//                           new row/column comes out of the same rank-1 form), the initial inverse is a Gauss-Jordan
//                           sweep whose only LDS traffic is the broadcast of the pivot column.
//   global mode (|W| <= 64): when the set outgrows 32 the wave copies Sinv to its slot of a global scratch and
//                           carries on (no restart); slower per update, but only the updates beyond 32 pay for it.
//
// The kernel is latency bound (one dependent chain per instance), so every active-set change is O(1) round trips:
//   add j   : c = G[W,j], u = Sinv c, sc = G_jj - c'u, mu = (t_j - b_j)/sc;  lam_W -= u mu, lam_j = mu;
//             t -= mu (G[:,j] - G[:,W] u);  Sinv bordered.
//   drop p  : s = Sinv[:,p], a = lam_p / s_p;  lam -= s a;  t += a G[:,W] s;  Sinv down-dated.
// where t = v0 - G[:,W] lam is the minimiser of the current face (kept for every row) -- algebraically the same
// iterates as recomputing lam and t from scratch (what oracle/mpc_oracle.py::polish_active_set does).  Before
// accepting a face as optimal lam and t ARE recomputed from scratch and the tests repeated, so rounding drift of
// the incremental form cannot end the loop early.  Reductions use the DPP path.
template <bool GLB>
struct PolishMode {
    static constexpr bool glb = GLB;
    static constexpr bool half = !GLB;
    static constexpr int WL = GLB ? 64 : 32;  // capacity and leading dimension
    static constexpr int PMASK = WL - 1;
};

// Rollout of one instance by one wave (fused tail of k_polish): Z[k] = [e_x[:,k]; e_u[:,k]] rows of length C = n+m in
// LDS; lane (i = lane / G, g = lane % G) holds CPL consecutive coefficients of row i of [A B] in registers, so a step
// is one short LDS read, CPL FMAs, a log2(G) butterfly and one LDS write.
template <int CPL>
__device__ __forceinline__ void rollout_steps(double* Z, int n, int m, int N, int G, int lane, const double* A,
                                              const double* B) {
    const int C = n + m;
    const int i = lane / G, g = lane % G;
    double coef[CPL];
    int jc[CPL];
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
        const int j = g * CPL + t;
        const bool ok = (i < n) && (j < C);
        const int ic = i < n ? i : 0;
        const double av = A[(size_t)(j < n ? j : 0) * n + ic], bv = B[(size_t)((j >= n && j < C) ? j - n : 0) * n + ic];
        coef[t] = ok ? (j < n ? av : bv) : 0.0;
        jc[t] = j < C ? j : C - 1;
    }
    for (int k = 0; k < N; ++k) {
        const double* zk = Z + (size_t)k * C;
        double zv[CPL];
#pragma unroll
        for (int t = 0; t < CPL; ++t) zv[t] = zk[jc[t]];
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < CPL; ++t) acc += coef[t] * zv[t];
        if (G == 4) {  // butterfly inside each quad on the DPP path
            acc += dpp_d<0xB1, 0xF>(acc);  // quad_perm [1,0,3,2]
            acc += dpp_d<0x4E, 0xF>(acc);  // quad_perm [2,3,0,1]
        } else {
            for (int o = 1; o < G; o <<= 1) acc += __shfl_xor(acc, o);
        }
        if (g == 0 && i < n) Z[(size_t)(k + 1) * C + i] = acc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// Blocked rollout of one instance by one wave (shared model).  The recursion e+ = A e + B v is a chain of N dependent steps, each an
// LDS (or cross-lane) round trip; here s = floor(64 / n) stages advance at once: lane (j, i) owns row i of stage k0 + j + 1 and
//   e_x[k0+j+1] = A^(j+1) e_x[k0] + sum_{t<=j} A^(j-t) B e_u[k0+t],
// with its row of [Gamma_s | Phi_s] in registers (rollM, built at design time).  The input part does not depend on the chain and is
// summed while the previous block's values leave; the block's start state reaches every lane through v_readlane (SGPR operands of
// the FMAs): no LDS on the chain.  Values leave for HBM as they are produced (lane order = memory order of x and e_x), so there is
// no trajectory buffer.  eub: LDS, e_u of the instance in stage order, ZERO beyond nz (256 doubles).
constexpr int ROLL_SMX = 24, ROLL_NX = 16;  // blocked rollout: s*m <= ROLL_SMX input columns, n <= ROLL_NX state columns per lane
// rollM is [ROLL_SMX + ROLL_NX][64], zero beyond the s*m input / n state columns in use and on lanes >= s*n: every loop below is
// straight-line code of fixed length (no per-column branches, all LDS reads of a block in flight together).  <SMX, NX> are the
// column counts the instantiation carries in registers (the benchmark shape has its exact fit <20, 12>).
template <int SMX, int NX>
__device__ __forceinline__ void roll_load(const double* M, int lane, double (&cu)[SMX], double (&cx)[NX]) {
#pragma unroll
    for (int c = 0; c < SMX; ++c) cu[c] = M[c * 64 + lane];
#pragma unroll
    for (int c = 0; c < NX; ++c) cx[c] = M[(ROLL_SMX + c) * 64 + lane];
}
template <int SMX, int NX>
__device__ __forceinline__ void roll_run(const double (&cu)[SMX], const double (&cx)[NX], int n, int m, int N, int s, int nb, int lane,
                                         const double* eub, double e0v, double x0r, const double* xref_lds, const double* xref_glb,
                                         double* gx, double* gex) {
    const int nx = n * (N + 1), sm = s * m;
    if (lane < n) {  // stage 1 of the reference's numbering: x[:,1] = x0, e_x[:,1] = x0 - x_ref[:,1]
        gex[lane] = e0v;
        if (gx) gx[lane] = x0r;   // (gx == nullptr: only e_x is wanted -- the state-row finish rolls v0 out into LDS)
    }
    const bool mine = lane < s * n;
    // input part of a block: independent of the chain, so the next block's is summed while this block's values leave
    auto upart = [&](int b, double (&acc)[4]) {
        const double* ub = eub + b * sm;
        double uv[SMX];
#pragma unroll
        for (int c = 0; c < SMX; ++c) uv[c] = ub[c];   // zero padded: reads past the block's inputs meet zero coefficients
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = 0.0;
#pragma unroll
        for (int c = 0; c < SMX; ++c) acc[c & 3] = __builtin_fma(cu[c], uv[c], acc[c & 3]);
    };
    double acc[4];
    upart(0, acc);
    double prev = e0v;  // the block's start state: row i on lane src0 + i
    int src0 = 0;
#pragma unroll 1
    for (int b = 0; b < nb; ++b) {
        const int t = (b * s + 1) * n + lane;       // flat index of this lane's value in x / e_x
        const bool ok = mine && t < nx;
        const int tc = ok ? t : 0;
        double xr;                                   // reference: LDS copy of a shared one, or global per instance
        if (xref_lds) xr = xref_lds[tc];
        else xr = xref_glb[tc];
#pragma unroll
        for (int c = 0; c < NX; ++c) {
            const int sl = src0 + c;
            acc[c & 3] = __builtin_fma(cx[c], readlane_d(prev, sl < 64 ? sl : 63), acc[c & 3]);  // columns >= n: zero coefficient
        }
        const double ev = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        prev = ev;
        src0 = (s - 1) * n;
        upart(b + 1, acc);                           // (one block past the end reads the zero padding)
        if (xref_lds) asm volatile("" : "+v"(xr));   // keeps the two reference paths apart (a select of the addresses = flat loads)
        if (ok) {
            gex[t] = ev;
            if (gx) gx[t] = ev + xr;
        }
    }
}

// Two builds of the polish.  k_polish<false>: 4 waves per workgroup, one instance per wave, G read through L2.
// k_polish<true>: ONE persistent workgroup of 8 waves per CU that first copies the shared G = H'^-1 into LDS (115 KB
// for nz = 120: it fits beside the waves' small buffers in the 160 KB of a gfx950 CU) and then lets each wave pull
// instances from a queue (hard ones first, see k_admm) until it is empty: every row of G an update needs is an LDS
// read instead of an L2 round trip, which is what the dependent chain of an active-set change was waiting on.
constexpr int POLISH_WAVES = 4;
#ifndef ALMPC_EXP_POLISH_WAVES
#define ALMPC_EXP_POLISH_WAVES 8
#endif
constexpr int POLISH_WAVES_GLDS = ALMPC_EXP_POLISH_WAVES;   // (experiments: -DALMPC_EXP_POLISH_WAVES=12|16 on the two-kernel path)
// LDS per wave (doubles): row buffer 128 | two position buffers 64 | row-index buffer (64 ints); the trajectory
// buffer of the fused rollout lies over the same words (the active-set state is dead by then)
constexpr int POLISH_LDS_MIN_PER_WAVE = 128 + 64 + 64 + 32;
// Workgroup-shared constants kept in LDS (doubles): d[nzs] | umin[m] | umax[m] | u_ref[nz] and, with the fused rollout,
// [A B] (n x (n+m), column-major) | x_ref[(N+1) n].  The two references are only used from here when they are shared by
// all instances (stride 0); per-instance references are read from global memory as before.
struct PolishShared {
    int off_d, off_umin, off_umax, off_uref, off_ab, off_xref, total;
};
__host__ __device__ inline PolishShared polish_shared_layout(int n, int m, int N, int nz, int nzs, int fused) {
    PolishShared L;
    L.off_d = 0;
    L.off_umin = nzs;
    L.off_umax = L.off_umin + m;
    L.off_uref = L.off_umax + m;
    L.off_ab = L.off_uref + nz;
    L.off_xref = L.off_ab + ((fused && fused != 3) ? n * (n + m) : 0);   // the blocked rollout (3) has no use for [A B]
    L.total = (L.off_xref + (fused ? (N + 1) * n : 0) + 1) & ~1;
    return L;
}
constexpr int POLISH_GLB_PER_INST = 64 * 64;        // doubles of global scratch per instance
constexpr int POLISH_SG_SHARED_CAP = 48;            // capacity (positions) of the workgroup-shared second-tier slot in LDS

typedef double d2 __attribute__((ext_vector_type(2)));

// GLDS: G in LDS (persistent 8-wave workgroups, tile-local queue); GPRE: it is there already (fused step kernel: requested
// before the ADMM phase).  KOFF: byte offset of the PolishParams inside the kernel-argument segment.
// SGL (per-instance models, small batches, e.g. the SQP loop where every instance has about 50 active rows): single-wave
// workgroups whose LDS holds the instance's own G_i (requested with direct global -> LDS loads at the start of the instance) and
// the 64 x 64 Sinv of a working set beyond 32 rows -- a bordering step beyond 32 rows is then two sweeps over LDS instead of
// one over global scratch and one over rows of G_i in L2.
template <bool GLDS, bool GPRE, int KOFF, bool SGL = false, bool START = false>
__device__ __forceinline__ void polish_body(const PolishParams& p_arg, double* smem) {
    const PolishParams& p = p_arg;
    constexpr int CH = (GLDS || SGL) ? 8 : 16;   // (16 for the LDS homes too: measured -6 % on the headline, round 3)  // positions per chunk of G rows (LDS latency needs fewer loads in flight than L2 latency)
    constexpr int NWV = GLDS ? POLISH_WAVES_GLDS : (SGL ? 1 : POLISH_WAVES);  // waves per workgroup
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane_k = threadIdx.x & 63;   // (wave-uniform, and the compiler knows it)
    const int nz = p.nz, nzs = p.nzs;
    const double* Gp = p.G;
    const int gs = GLDS ? ((nz + 1) & ~1) : nzs;  // row stride of G as the waves read it: compact in LDS, nzs in global memory
    const PolishShared SL = polish_shared_layout(p.roll.n, p.m, p.roll.N, nz, nzs, p.fuse_rollout);
    double* shc = smem + (GLDS ? (size_t)nz * gs : 0);  // workgroup-shared constants
    double* wave_lds = shc + SL.total + (size_t)wv * p.lds_per_wave;
    int* qcnt = nullptr;
    const bool uref_sh = p.uref_stride == 0, xref_sh = p.roll.xref_stride == 0;
    const bool per_inst = p.G_stride != 0;  // host: only with GLDS = false
    double* cd = shc + SL.off_d;    // d and [A B] as the waves read them: workgroup-shared, or the wave's own copy
    double* cab = shc + SL.off_ab;
    if (per_inst) { cd = wave_lds + p.wave_const_off; cab = cd + nzs; }
    {
        constexpr int TPB = 64 * NWV;
        const int n = p.roll.n, m = p.m, N = p.roll.N;
        if (!per_inst)
            for (int t = threadIdx.x; t < nzs; t += TPB) shc[SL.off_d + t] = p.dvec[t];
        for (int t = threadIdx.x; t < m; t += TPB) { shc[SL.off_umin + t] = p.umin[t]; shc[SL.off_umax + t] = p.umax[t]; }
        if (uref_sh)
            for (int t = threadIdx.x; t < nz; t += TPB) shc[SL.off_uref + t] = p.uref[t];
        if (p.fuse_rollout) {
            if (!per_inst && p.fuse_rollout != 3) {
                for (int t = threadIdx.x; t < n * n; t += TPB) shc[SL.off_ab + t] = p.roll.A[t];
                for (int t = threadIdx.x; t < n * m; t += TPB) shc[SL.off_ab + n * n + t] = p.roll.B[t];
            }
            if (xref_sh)
                for (int t = threadIdx.x; t < (N + 1) * n; t += TPB) shc[SL.off_xref + t] = p.roll.xref[t];
        }
    }
    ALMPC_STAMP(8192 + blockIdx.x * 8 + wv, 0);
    if constexpr (GLDS && !GPRE) {
        // G -> LDS, rows packed to stride gs: wave w copies rows w, w + 8, ...; lane l the 16-byte pair l of a row; 8 rows in
        // flight per wave and round (a plain copy loop pays one L2 round trip per row)
        const int hs = gs / 2;
        const bool lane_in = lane_k < hs;
        const int lq = lane_in ? lane_k : 0;
        for (int r0_ = wv; r0_ < nz; r0_ += POLISH_WAVES_GLDS * 8) {
            d2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0_ + u * POLISH_WAVES_GLDS;
                v[u] = *reinterpret_cast<const d2*>(p.G + (size_t)(r < nz ? r : 0) * nzs + 2 * lq);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0_ + u * POLISH_WAVES_GLDS;
                if (r < nz && lane_in) *reinterpret_cast<d2*>(smem + (size_t)r * gs + 2 * lq) = v[u];
            }
        }
    }
    if constexpr (GLDS) {
        qcnt = reinterpret_cast<int*>(shc + SL.total + (size_t)POLISH_WAVES_GLDS * p.lds_per_wave);
        if (threadIdx.x == 0) { qcnt[0] = POLISH_WAVES_GLDS; qcnt[1] = -1; }  // queue counter | owner of the shared second-tier slot
        Gp = smem;
    }
    __syncthreads();
  auto process = [&](const int inst) {
    // the parameters are re-read from the kernarg segment through a pointer the optimiser cannot see through: otherwise
    // every instance-independent load (bounds, scaling, [A B] coefficients, ~40 pointers) is hoisted out of the queue
    // loop and the kernel spills hundreds of registers
    auto ka_ = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka_));
    const PolishParams& p = *(const PolishParams*)((const char*)ka_ + KOFF);
    int lane = lane_k;
    asm volatile("" : "+v"(lane));  // same reason: masks and constants derived from the lane number stay inside the instance
    ALMPC_STAMP(inst, 8);
    if constexpr (!GLDS) {
        if (per_inst) {  // this instance's G, d, [A B]
            if constexpr (SGL) {
                // G_i -> LDS, one row per wave instruction (lanes x 16 bytes), no registers: everything is in flight at once
                double* gl = wave_lds + p.g_off;
                const char* src = reinterpret_cast<const char*>(GL(p.G) + (size_t)inst * p.G_stride) + lane * 16;
                if (lane < nzs / 2)
                    for (int r = 0; r < nz; ++r)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)r * nzs * 8),
                                                         (__attribute__((address_space(3))) void*)(gl + (size_t)r * nzs), 16, 0, 0);
                Gp = gl;
            } else
            Gp = GL(p.G) + (size_t)inst * p.G_stride;
            const int n_ = p.roll.n, m_ = p.m;
            for (int t = lane; t < nzs; t += 64) cd[t] = GL(p.dvec)[(size_t)inst * p.d_stride + t];
            if (p.fuse_rollout) {
                for (int t = lane; t < n_ * n_; t += 64) cab[t] = GL(p.roll.A)[(size_t)inst * p.A_stride + t];
                for (int t = lane; t < n_ * m_; t += 64) cab[n_ * n_ + t] = GL(p.roll.B)[(size_t)inst * p.B_stride + t];
            }
            if constexpr (SGL) __builtin_amdgcn_s_waitcnt(0);  // the copy of G_i has landed (vmcnt and lgkmcnt zero)
            wave_fence_lds();
        }
    }
    const int st_in = GL(p.status)[inst];
    const size_t base = (size_t)inst * nzs;
    // row-distributed vectors: lane l owns the two consecutive rows 2l, 2l+1 (one 16-byte access per vector)
    const int r0 = 2 * lane, r1 = 2 * lane + 1;
    const bool in0 = r0 < nz, in1 = r1 < nz;
    const bool inrow = r0 < nzs;            // nzs is even: the pair (r0, r1) is inside the padded vector or not at all
    const int rc = inrow ? r0 : 0;          // clamped pair index for loads (lanes beyond the vector read pair 0)
    const bool skip = (st_in == 2);  // non-finite instance: nothing to polish, the ADMM iterate is handed on as is

    double* rowbuf = wave_lds;         // [128] one row-distributed vector, for gathers by row index
    double* pbufa = rowbuf + 128;      // [64]  one position-distributed vector, for broadcasts by position
    double* pbufb = pbufa + 64;        // [64]  a second one
    int* wrow_s = reinterpret_cast<int*>(pbufb + 64);  // [64] row index of each position (copy of wrow)
    // second tier (working sets beyond 32 rows): Sinv in memory, leading dimension 64, capacity cap2.  Its home is the wave's own
    // LDS (SGL), the workgroup-shared LDS slot if this wave gets it (G-in-LDS builds; chosen when the tier is entered), or the
    // instance's global scratch.  In the G-in-LDS builds the pointer is a generic one (flat accesses): the tier is rare there.
    double* Sg;
    int cap2 = 64;
    bool own_slot = false;
    // (single-wave builds keep it in the wave's LDS unless the host gave it no room there: sg_off < 0 -- k_step_inst_wave, where a
    // 32 KB slot per wave would leave three waves per CU for four instances)
    const bool sg_lds = SGL && p.sg_off >= 0;
    if (sg_lds) Sg = wave_lds + p.sg_off;
    else Sg = GL(p.sglobal) + (size_t)inst * POLISH_GLB_PER_INST;

    double lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0, v00, v01, w0 = 0, w1 = 0, y0, y1, z0, z1;
    // x0 of this instance for the fused rollout (n <= 64: one state per lane), requested with the other prologue loads
    double x0r = 0.0;
    if (p.fuse_rollout && lane < p.roll.n) x0r = GL(p.roll.x0)[(size_t)inst * p.roll.n + lane];
    {
        const d2 dv = *reinterpret_cast<const d2*>(cd + rc);
        const d2 vv = *reinterpret_cast<const d2*>(GL(p.v0) + base + rc);
        d2 yy;
        if (p.yflags) {  // only the signs of y are needed (the guess below): +-1 / 0 from the flag word of this row pair
            const uint32_t wd = GL(p.yflags)[(size_t)inst * p.yflag_words + (rc >> 4)];
            const uint32_t b = (uint32_t)(rc & 15);
            yy[0] = (double)(int)((wd >> (16u + b)) & 1u) - (double)(int)((wd >> b) & 1u);
            yy[1] = (double)(int)((wd >> (17u + b)) & 1u) - (double)(int)((wd >> (b + 1u)) & 1u);
        } else {
            yy = *reinterpret_cast<const d2*>(GL(p.ys) + base + rc);
        }
        const d2 zz = *reinterpret_cast<const d2*>(GL(p.zs) + base + rc);
        v00 = vv[0]; v01 = vv[1]; y0 = yy[0]; y1 = yy[1]; z0 = zz[0]; z1 = zz[1];
        // bounds exactly as k_admm forms them ((umin - uref) * (1/d)): its z sits ON these values when active
        // (references: LDS copy of a shared one, or global per instance.  Two code paths kept apart by an empty asm: a select of the
        // two addresses -- which is what the optimiser makes of a plain if/else as well -- is a flat load that waits for every
        // store in flight, here the trajectory stores of the previous instance)
        double urp0, urp1;
        if (uref_sh) {
            urp0 = shc[SL.off_uref + (in0 ? r0 : 0)]; urp1 = shc[SL.off_uref + (in1 ? r1 : 0)];
            asm volatile("" : "+v"(urp0), "+v"(urp1));
        } else {
            const double* ug = GL(p.uref) + (size_t)inst * p.uref_stride;
            urp0 = ug[in0 ? r0 : 0]; urp1 = ug[in1 ? r1 : 0];
        }
        if (in0) {
            const double di = 1.0 / dv[0], ur = urp0;
            lo0 = (shc[SL.off_umin + r0 % p.m] - ur) * di; hi0 = (shc[SL.off_umax + r0 % p.m] - ur) * di; w0 = fmin(fmax(z0, lo0), hi0);
        } else { v00 = 0.0; y0 = 0.0; z0 = 0.0; }
        if (in1) {
            const double di = 1.0 / dv[1], ur = urp1;
            lo1 = (shc[SL.off_umin + r1 % p.m] - ur) * di; hi1 = (shc[SL.off_umax + r1 % p.m] - ur) * di; w1 = fmin(fmax(z1, lo1), hi1);
        } else { v01 = 0.0; y1 = 0.0; z1 = 0.0; }
    }
    int wrow = 0, wsd = 0;   // position-distributed: row index, side (+1 upper / -1 lower)
    double wbnd = 0.0;       // position-distributed: bound value
    double lam = 0.0;        // position-distributed: multiplier of the bound (H't + f' + E_W lam = 0)
    bool act0 = false, act1 = false;  // row-distributed: row is in the working set
    double bnd0 = 0.0, bnd1 = 0.0;    // row-distributed: the bound it sits on
    double t0 = v00, t1 = v01;        // row-distributed: face minimiser v0 - G[:,W] lam
    int k = 0;               // |W|, wave-uniform
    bool overflow = false;   // the set outgrew the current mode
    ALMPC_ACC_DECL
    wrow_s[lane] = 0;
    const int hpos = lane & 31, hhf = lane >> 5;
    double Sr[16];  // register mode: Sr[t] = Sinv[hpos][16 hhf + t], exactly zero outside the k x k block
#pragma unroll
    for (int t = 0; t < 16; ++t) Sr[t] = 0.0;
    auto bcast16 = [&](const double* buf, double (&o)[16]) {  // this half's 16 entries of a position buffer
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const d2 v = *reinterpret_cast<const d2*>(buf + 16 * hhf + 2 * t);
            o[2 * t] = v[0]; o[2 * t + 1] = v[1];
        }
    };
    auto col_to_buf = [&](int j, double* buf) {  // buf[pos] = Sinv[pos][j] for a wave-uniform j (register mode)
        const int tj = j & 15;
        double v = Sr[0];
#pragma unroll
        for (int t = 1; t < 16; ++t) {
            double x = Sr[t];
            asm volatile("" : "+v"(x));  // keeps this a select of VALUES: a select of addresses would put Sr in scratch
            v = (tj == t) ? x : v;
        }
        if (hhf == (j >> 4)) buf[hpos] = v;
        wave_fence_lds();
    };
    ALMPC_STAMP(inst, 9);

    // Sinv is padded beyond the k x k block (zeros in register mode, the identity in global mode) and
    // position-distributed vectors are exactly zero beyond position k, so every sweep below runs over chunk-rounded
    // ranges with UNCONDITIONAL loads and stores (no exec-masked branches around memory operations: they would
    // serialise every LDS round trip).
    // Broadcasts of a position-distributed vector go through a 64-entry LDS buffer (one uniform-address read per
    // element, pipelined with the Sinv reads) rather than through v_readlane pairs.
    auto sync_s = [&](auto m) {  // order this wave's writes to Sinv before its later reads
        if (decltype(m)::glb && !sg_lds) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            wave_fence_lds();
        }
    };
    auto sptr = [&](auto) -> double* { return Sg; };  // only the global mode keeps Sinv in memory
    auto put_pos = [&](auto m, double* buf, double v) {  // position-distributed register -> LDS buffer
        buf[decltype(m)::half ? (lane & 31) : lane] = v;  // (both halves write the same value in LDS mode)
        wave_fence_lds();
    };

    // u = Sinv * c, c given in LDS buffer cb (zero beyond k); result position-distributed (zero beyond k)
    auto s_matvec = [&](auto m, const double* cb) -> double {
        using M = decltype(m);
        double acc = 0.0;
        if constexpr (M::half) {
            double cv[16];
            bcast16(cb, cv);
            double acc2 = 0.0;
#pragma unroll
            for (int t = 0; t < 16; t += 2) { acc += Sr[t] * cv[t]; acc2 += Sr[t + 1] * cv[t + 1]; }
            acc = half_sum(acc + acc2);
        } else {
            double* S = sptr(m);
            for (int l0 = 0; l0 < k; l0 += 8) {
                double sv[8], cv[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) { sv[t] = S[(l0 + t) * M::WL + lane]; cv[t] = cb[l0 + t]; }
#pragma unroll
                for (int t = 0; t < 8; ++t) acc += sv[t] * cv[t];
            }
        }
        return acc;
    };
    // Sinv += a a' * scale: a position-distributed in register AND in LDS buffer ab (zero beyond k)
    auto s_rank1 = [&](auto m, double a, const double* ab, double scale) {
        using M = decltype(m);
        const double as = a * scale;
        if constexpr (M::half) {
            double av[16];
            bcast16(ab, av);
#pragma unroll
            for (int t = 0; t < 16; ++t) Sr[t] = __builtin_fma(as, av[t], Sr[t]);
        } else {
            double* S = sptr(m);
            for (int l0 = 0; l0 < k; l0 += 8) {
                double cur[8], av[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) { cur[t] = S[(l0 + t) * M::WL + lane]; av[t] = ab[l0 + t]; }
#pragma unroll
                for (int t = 0; t < 8; ++t) S[(l0 + t) * M::WL + lane] = cur[t] + as * av[t];
            }
        }
    };
    // (q0, q1) = sum_{l<k} G[W_l, rows] * a_l, a given in LDS buffer ab (zero beyond k): the only O(nz k) part of an
    // update.  One 16-byte load per row and lane; row indices come from the LDS copy of wrow (zero beyond k: row 0
    // is a valid address and its weight is zero).  Split in a load half and an FMA half so that the first chunk's L2
    // round trip can be started before the LDS work that produces the weights.
    auto g_load = [&](int l0, d2 (&g)[CH]) {
        // row indices straight from the position-distributed register (v_readlane -> scalar address part): the LDS copy wrow_s cost a
        // dependent LDS round trip per chunk before the rows of G could even be requested
#ifdef ALMPC_EXP_ROWS_READLANE
#pragma unroll
        for (int t = 0; t < CH; ++t) g[t] = *reinterpret_cast<const d2*>(Gp + (__builtin_amdgcn_readlane(wrow, l0 + t) * gs + rc));
#else
        // Round 3: CH row indices in CH / 4 wave-uniform 16-byte reads of the LDS copy, addresses on the vector unit.  v_readlane ->
        // s_mul -> address costs ~30 cycles per row one after the other (tools/microbench/dep_latency.hip: a VALU write to a scalar
        // register is slow to reach its reader), 1.0 k of the 4.1 k cycles of a one-row change at 30 rows; the LDS copy is ONE ~80-cycle
        // round trip per chunk.  (What round 2 measured as slower was one dependent ds_read per row.)
        typedef int i4 __attribute__((ext_vector_type(4)));
        int rows[CH];
#pragma unroll
        for (int t = 0; t < CH; t += 4) {
            const i4 r4 = *reinterpret_cast<const i4*>(wrow_s + l0 + t);
            rows[t] = r4[0]; rows[t + 1] = r4[1]; rows[t + 2] = r4[2]; rows[t + 3] = r4[3];
        }
#pragma unroll
        for (int t = 0; t < CH; ++t) g[t] = *reinterpret_cast<const d2*>(Gp + (rows[t] * gs + rc));
#endif
    };
    auto g_fma = [&](int l0, const double* ab, const d2 (&g)[CH], double& q0, double& q1) {
        double av[CH];
#pragma unroll
        for (int t = 0; t < CH; ++t) av[t] = ab[l0 + t];
#pragma unroll
        for (int t = 0; t < CH; ++t) { q0 += g[t][0] * av[t]; q1 += g[t][1] * av[t]; }
    };
    auto g_rows_times = [&](const double* ab, double& q0, double& q1) {
        q0 = 0.0; q1 = 0.0;
        for (int l0 = 0; l0 < k; l0 += CH) {
            d2 g[CH];
            g_load(l0, g);
            g_fma(l0, ab, g, q0, q1);
        }
    };
    auto put_rows = [&](double a0, double a1) {  // row-distributed pair -> rowbuf
        d2 v; v[0] = a0; v[1] = a1;
        *reinterpret_cast<d2*>(rowbuf + r0) = v;
        wave_fence_lds();
    };
    auto recompute = [&](auto m) {  // lam = Sinv (v0_W - b), t = v0 - G[:,W] lam from scratch
        using M = decltype(m);
        const int pos = M::half ? (lane & 31) : lane;
        put_rows(v00, v01);
        const double rv = rowbuf[wrow] - wbnd;  // wrow is always a valid row index
        put_pos(m, pbufa, (pos < k) ? rv : 0.0);
        const double lm = s_matvec(m, pbufa);
        lam = (pos < k) ? lm : 0.0;
        put_pos(m, pbufb, lam);
        double q0, q1;
        g_rows_times(pbufb, q0, q1);
        t0 = v00 - q0;
        t1 = v01 - q1;
    };
    // add row j (uniform) at bound bval with side sd; the caller guarantees k < capacity
    auto add_row = [&](auto m, int j, double bval, int sd) {
        using M = decltype(m);
        const int pos = M::half ? (lane & 31) : lane;
        const bool lowhalf = M::half ? (lane < 32) : true;
        ALMPC_ACC_START;
        const d2 gj = *reinterpret_cast<const d2*>(Gp + (j * gs + rc));  // row j = column j
        d2 g[CH];
        g_load(0, g);  // rows of positions 0..CH-1: in flight while Sinv c is formed
        // c = G[W, j] = G[j, W] (symmetric) gathered straight from row j, beside the load of the row itself
        const double cv = Gp[j * gs + wrow];  // wrow is always a valid row index
        const double gjj = Gp[j * gs + j];
        const double c = (pos < k) ? cv : 0.0;
        const double tj = readlane_d((j & 1) ? t1 : t0, j >> 1);
        ALMPC_ACC(0);
        put_pos(m, pbufa, c);
        const double u = s_matvec(m, pbufa);  // zero beyond k (identity padding)
        ALMPC_ACC(1);
        put_pos(m, pbufb, u);
        double q0 = 0.0, q1 = 0.0;
        g_fma(0, pbufb, g, q0, q1);
        for (int l0 = CH; l0 < k; l0 += CH) {
            g_load(l0, g);
            g_fma(l0, pbufb, g, q0, q1);
        }
        // (measured and dropped, round 3: two chunks in flight -- the loads of chunk c + 1 ahead of the FMAs of chunk c: mixed batch -1 %)
        ALMPC_ACC(2);
        const double sc = gjj - wave_sum(lowhalf ? c * u : 0.0);
        const double isc = fast_rcp_d(sc);
        const double mu = (tj - bval) * isc;
        t0 -= mu * (gj[0] - q0);
        t1 -= mu * (gj[1] - q1);
        lam -= u * mu;
        ALMPC_ACC(3);
        if constexpr (M::half) {
            // [Sinv + u u'/sc, -u/sc; -u'/sc, 1/sc] as ONE rank-1 update of the zero-padded registers: with -1 in
            // position k of the broadcast vector and -1/sc as the factor of lane k, the new row, column and corner come
            // out of the same FMA as the update of the old block
            if (lane == k) {
                double m1 = -1.0;
                asm volatile("" : "+v"(m1));  // materialised here (the optimiser otherwise keeps it in a spill slot)
                pbufb[hpos] = m1;  // hpos == k on this lane: the address put_pos already uses (k < 32 in register mode)
            }
            wave_fence_lds();
            double av[16];
            bcast16(pbufb, av);
            const double as = (pos == k) ? -isc : u * isc;
#pragma unroll
            for (int t = 0; t < 16; ++t) Sr[t] = __builtin_fma(as, av[t], Sr[t]);
        } else {
            double* S = sptr(m);
            s_rank1(m, u, pbufb, isc);
            // border: new column k and new row k (u is zero beyond k, so the padding stays zero)
            const double bv = (pos == k) ? isc : -u * isc;
            S[k * M::WL + pos] = bv;
            if (pos < cap2) S[pos * M::WL + k] = bv;
        }
        if (pos == k) { wrow = j; wsd = sd; wbnd = bval; lam = mu; }
        if (lane == 0) wrow_s[k] = j;
        if (r0 == j) { act0 = true; bnd0 = bval; }
        if (r1 == j) { act1 = true; bnd1 = bval; }
        k += 1;
        sync_s(m);
        ALMPC_ACC(4);
#ifdef ALMPC_STAMPS
        acc_n += 1;
#endif
    };
    // remove position rp (uniform)
    auto remove_pos = [&](auto m, int rp) {
        using M = decltype(m);
        const int pos = M::half ? (lane & 31) : lane;
        const int last = k - 1;
        const int jrem = __builtin_amdgcn_readlane(wrow, rp);
        if constexpr (M::half) {
            col_to_buf(rp, pbufb);
            const double sp = pbufb[pos];  // column rp of Sinv (zero beyond k)
            double av[16];
            bcast16(pbufb, av);
            const double spp = readlane_d(sp, rp);
            const double a = readlane_d(lam, rp) / spp;
            double q0, q1;
            g_rows_times(pbufb, q0, q1);
            t0 += a * q0;
            t1 += a * q1;
            lam -= sp * a;
            const double as = -sp / spp;
#pragma unroll
            for (int t = 0; t < 16; ++t) Sr[t] = __builtin_fma(as, av[t], Sr[t]);  // row and column rp become ~zero
            if (rp != last) {  // move the last position into the hole
                col_to_buf(last, pbufa);
                const double colv = pbufa[pos];
                const double corner = readlane_d(colv, last);
                const double nv = (pos == rp) ? corner : ((pos == last) ? 0.0 : colv);
                put_pos(m, pbufa, nv);
                double rv[16];
                bcast16(pbufa, rv);
#pragma unroll
                for (int t = 0; t < 16; ++t) Sr[t] = (pos == rp) ? rv[t] : ((16 * hhf + t == rp) ? nv : Sr[t]);
                const int lrow = __builtin_amdgcn_readlane(wrow, last), lsd = __builtin_amdgcn_readlane(wsd, last);
                const double lbv = readlane_d(wbnd, last), llam = readlane_d(lam, last);
                if (pos == rp) { wrow = lrow; wsd = lsd; wbnd = lbv; lam = llam; }
                if (lane == 0) wrow_s[rp] = lrow;
            }
            // position `last` returns to the zero padding
#pragma unroll
            for (int t = 0; t < 16; ++t) Sr[t] = (pos == last || 16 * hhf + t == last) ? 0.0 : Sr[t];
        } else {
            double* S = sptr(m);
            const double sp = S[rp * M::WL + pos];  // column rp of Sinv (zero beyond k)
            const double spp = readlane_d(sp, rp);
            const double a = readlane_d(lam, rp) / spp;
            put_pos(m, pbufb, sp);
            double q0, q1;
            g_rows_times(pbufb, q0, q1);
            t0 += a * q0;
            t1 += a * q1;
            lam -= sp * a;
            s_rank1(m, sp, pbufb, -1.0 / spp);  // row and column rp become zero
            sync_s(m);
            if (rp != last) {  // move the last position into the hole
                const double colv = S[last * M::WL + pos];
                const double corner = readlane_d(colv, last);
                const double nv = (pos == rp) ? corner : ((pos == last) ? 0.0 : colv);
                S[rp * M::WL + pos] = nv;
                if (pos < cap2) S[pos * M::WL + rp] = nv;
                const int lrow = __builtin_amdgcn_readlane(wrow, last), lsd = __builtin_amdgcn_readlane(wsd, last);
                const double lbv = readlane_d(wbnd, last), llam = readlane_d(lam, last);
                if (pos == rp) { wrow = lrow; wsd = lsd; wbnd = lbv; lam = llam; }
                if (lane == 0) wrow_s[rp] = lrow;
            }
            {   // position `last` returns to the identity padding
                const double iv = (pos == last) ? 1.0 : 0.0;
                S[last * M::WL + pos] = iv;
                if (pos < cap2) S[pos * M::WL + last] = iv;
            }
        }
        if (pos == last) lam = 0.0;
        if (lane == 0) wrow_s[last] = 0;
        if (r0 == jrem) act0 = false;
        if (r1 == jrem) act1 = false;
        k -= 1;
        sync_s(m);
    };

    int it = 0;
    int fin = 1;         // 0 = certified
    bool fresh = true;   // lam and t were recomputed from scratch since the last change of W
    const int max_iter = p.max_iter;
    // the active-set loop in one storage mode; returns with overflow = true if an add does not fit
    auto run = [&](auto m) {
        using M = decltype(m);
        const int pos = M::half ? (lane & 31) : lane;
        const bool lowhalf = M::half ? (lane < 32) : true;
        while (it < max_iter) {
            ++it;
            ALMPC_ACC_START;
            // ---- ratio test over the free rows: first bound hit on the way from w to the face minimiser t
            double rr = __builtin_inf();
            int rside = 0;
            bool second = false;
            {
                const bool f0 = in0 && !act0, f1 = in1 && !act1;
                const bool up0 = t0 > hi0, dn0 = t0 < lo0, up1 = t1 > hi1, dn1 = t1 < lo1;
                const double c0 = ((up0 ? hi0 : lo0) - w0) * fast_rcp_d(t0 - w0);   // (only rows with t outside their box count: t != w there)
                const double c1 = ((up1 ? hi1 : lo1) - w1) * fast_rcp_d(t1 - w1);
                const bool v0 = f0 && (up0 || dn0), v1 = f1 && (up1 || dn1);
                if (v0) { rr = c0; rside = up0 ? 1 : -1; }
                if (v1 && (!v0 || c1 < c0)) { rr = c1; rside = up1 ? 1 : -1; second = true; }  // ties keep the smaller row
            }
            const double rmin = wave_min(rr);
            if (rmin < 1.0) {
                if (k == (M::glb ? cap2 : M::WL)) { --it; overflow = true; return; }  // redo this pass in the next mode
                const int owner = __builtin_ctzll(__ballot(rr == rmin));  // smallest lane = smallest row among ties
                // every lane has its own candidate ready (row parity, side, bound): three v_readlane that depend on `owner` only,
                // instead of a chain owner -> parity -> select -> bound
                const double bcand = second ? (rside > 0 ? hi1 : lo1) : (rside > 0 ? hi0 : lo0);
                const int jmin = 2 * owner + (__builtin_amdgcn_readlane(second ? 1 : 0, owner));
                const int sd = __builtin_amdgcn_readlane(rside, owner);
                const double bval = readlane_d(bcand, owner);
                const double tt = fmax(rmin, 0.0);
                if (in0 && !act0) w0 += tt * (t0 - w0);
                if (in1 && !act1) w1 += tt * (t1 - w1);
                if (r0 == jmin) w0 = bval;
                if (r1 == jmin) w1 = bval;
                ALMPC_ACC(5);
                add_row(m, jmin, bval, sd);
                fresh = false;
                continue;
            }
            if (!act0) w0 = t0;
            if (!act1) w1 = t1;
            // ---- multiplier signs: upper bound needs lam >= 0, lower bound lam <= 0
            bool ok = true;
            int vi = 0;
            if (k > 0) {
                const bool mine = lowhalf && pos < k;
                const double viol = mine ? ((wsd > 0) ? -lam : lam) : -__builtin_inf();
                const double lmax = wave_max(mine ? fabs(lam) : 0.0);
                const double vmax = wave_max(viol);
                ok = vmax <= 1e-12 * fmax(1.0, lmax);
                vi = __builtin_ctzll(__ballot(mine && viol == vmax));  // smallest position among ties
            }
            if (ok) {
                if (fresh) { fin = 0; return; }
                recompute(m);  // confirm on values computed from scratch (does not count as an iteration)
                fresh = true;
                --it;
                continue;
            }
            remove_pos(m, vi);
            fresh = false;
        }
    };
    // entering the second tier with `need` positions wanted right away: pick its home
    auto tier2_home = [&](int need) {
        if constexpr (GLDS) {
            if (p.sg_shared_off >= 0 && need <= POLISH_SG_SHARED_CAP) {
                int got = 0;
                if (lane == 0) got = (atomicCAS(qcnt + 1, -1, wv) == -1) ? 1 : 0;
                got = __builtin_amdgcn_readfirstlane(got);
                if (got) { own_slot = true; cap2 = POLISH_SG_SHARED_CAP; Sg = shc + p.sg_shared_off; }
            }
        }
    };
    // the shared slot is full (cap2 positions): carry on in the global scratch (capacity 64)
    auto slot_to_global = [&]() {
        double* Gs = GL(p.sglobal) + (size_t)inst * POLISH_GLB_PER_INST;
        for (int c = 0; c < POLISH_SG_SHARED_CAP; c += 8) {
            double v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = Sg[(c + t) * 64 + lane];
#pragma unroll
            for (int t = 0; t < 8; ++t) Gs[(c + t) * 64 + lane] = v[t];
        }
        for (int c = POLISH_SG_SHARED_CAP; c < 64; ++c) Gs[c * 64 + lane] = (c == lane) ? 1.0 : 0.0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane == 0) atomicExch(qcnt + 1, -1);
        own_slot = false; cap2 = 64; Sg = Gs;
    };
    auto to_global = [&]() {  // carry Sinv (k = 32 positions) over to the second tier: leading dimension 64, identity padded
#pragma unroll
        for (int t = 0; t < 16; ++t) {  // Sinv is symmetric: lane (hpos, hhf) writes its piece of columns 16 hhf + t
            Sg[(16 * hhf + t) * 64 + hpos] = Sr[t];
            Sg[(16 * hhf + t) * 64 + 32 + hpos] = 0.0;
        }
#pragma unroll 4
        for (int c = 32; c < cap2; ++c) Sg[c * 64 + lane] = 0.0;
        if (lane >= 32 && lane < cap2) Sg[lane * 64 + lane] = 1.0;  // same lane as the zero above: program order
        if (lane >= 32) lam = 0.0;  // lanes 32..63 stop mirroring positions 0..31: they are positions 32..63 now
        sync_s(PolishMode<true>{});
    };

    // ---- initial working set from the ADMM multipliers (OSQP polish rule: sign of y), rows in ascending order.
    // Sinv of the first (up to 32) rows at once: gather G_WW into LDS, Gauss-Jordan in place (SPD: no pivoting);
    // flagged rows beyond 32 are then added one by one in global mode.
    const int s0 = (in0 && y0 < 0.0 && w0 <= lo0) ? -1 : ((in0 && y0 > 0.0 && w0 >= hi0) ? 1 : 0);
    const int s1 = (in1 && y1 < 0.0 && w1 <= lo1) ? -1 : ((in1 && y1 > 0.0 && w1 >= hi1) ? 1 : 0);
    const unsigned long long m0 = __ballot(s0 != 0), m1 = __ballot(s1 != 0);
    const int k0 = __popcll(m0) + __popcll(m1);
    bool give_up = skip || k0 > 64;  // more than 64 bounds active: beyond the largest mode, keep the ADMM iterate
    if (!give_up && k0 > 0) {
        using M = PolishMode<false>;
        const M m{};
        const int pos = lane & 31;
        // position of a flagged row = number of flagged rows before it (rows ascend with the lane, then first/second)
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const int p0 = __popcll(m0 & below) + __popcll(m1 & below), p1 = p0 + (s0 != 0 ? 1 : 0);
        int* ibuf = reinterpret_cast<int*>(rowbuf);  // [0..64): row of position, [64..128): side
        if (s0 != 0) { ibuf[p0] = r0; ibuf[64 + p0] = s0; }
        if (s1 != 0) { ibuf[p1] = r1; ibuf[64 + p1] = s1; }
        wave_fence_lds();
        bool started = false;
        if constexpr (START) {
            // A guessed set of 33..64 rows whose inverse was built ahead of this kernel (four waves per instance, in registers:
            // k_guess_iterate_ws): installed in the second tier as it stands.  Left to this kernel, the first 32 rows are one Gauss-Jordan
            // sweep in registers and every further row a bordering in memory mode, 6 k cycles each -- 108 k of the 190 k cycles of an
            // SQP iteration's finish (50 inputs of 100 on a bound).  The list is compared row by row: a guess this kernel reads differently
            // (a bound within rounding of the iterate) takes the ordinary route.
            if (p.start_rows && k0 > 32) {
                const int32_t* sr = GL(p.start_rows) + (size_t)inst * 65;
                const bool same = sr[0] == k0 && (lane >= k0 || sr[1 + lane] == ibuf[lane]);
                if (__all(same)) {
                    started = true;
                    k = k0;
                    wrow = lane < k0 ? ibuf[lane] : 0;
                    wsd = lane < k0 ? ibuf[64 + lane] : 0;
                    wave_fence_lds();
                    if (s0 != 0) { act0 = true; bnd0 = s0 > 0 ? hi0 : lo0; }
                    if (s1 != 0) { act1 = true; bnd1 = s1 > 0 ? hi1 : lo1; }
                    put_rows(s0 > 0 ? hi0 : lo0, s1 > 0 ? hi1 : lo1);
                    wbnd = lane < k0 ? rowbuf[wrow] : 0.0;
                    lam = 0.0;
                    wrow_s[lane] = wrow;
                    const double* src = GL(p.sglobal) + (size_t)inst * POLISH_GLB_PER_INST;
                    if (Sg != src) {
                        for (int c = 0; c < 64; c += 8) {
                            double v[8];
#pragma unroll
                            for (int t = 0; t < 8; ++t) v[t] = src[(c + t) * 64 + lane];
#pragma unroll
                            for (int t = 0; t < 8; ++t) Sg[(c + t) * 64 + lane] = v[t];
                        }
                    }
                    sync_s(PolishMode<true>{});
                    recompute(PolishMode<true>{});
                    overflow = true;   // continue in memory mode below
                }
            }
        }
        if (!started) {
        k = k0 < 32 ? k0 : 32;
        if (pos < k) { wrow = ibuf[pos]; wsd = ibuf[64 + pos]; }
        // pending rows (positions >= 32) are remembered on the lanes of the same number
        const int pend_row = (lane >= 32 && lane < k0) ? ibuf[lane] : 0;
        const int pend_sd = (lane >= 32 && lane < k0) ? ibuf[64 + lane] : 0;
        wave_fence_lds();
        if (s0 != 0 && p0 < 32) { act0 = true; bnd0 = s0 > 0 ? hi0 : lo0; }
        if (s1 != 0 && p1 < 32) { act1 = true; bnd1 = s1 > 0 ? hi1 : lo1; }
        put_rows(s0 > 0 ? hi0 : lo0, s1 > 0 ? hi1 : lo1);  // bound a flagged row sits on
        if (pos < k) wbnd = rowbuf[wrow];
        const double pend_bnd = (lane >= 32 && lane < k0) ? rowbuf[pend_row] : 0.0;
        // Two passes at most: if the multipliers of the guessed set come out with the wrong sign on two or more rows,
        // those rows are all dropped at once and the inverse is rebuilt for the rest (one more Gauss-Jordan sweep is
        // cheaper than one down-date per row; the primal active-set loop below starts from a feasible point and a
        // working set either way, so its finite termination does not depend on this shortcut).
        for (int pass = 0; pass < 2; ++pass) {
        if (lane < 32) wrow_s[lane] = (lane < k) ? wrow : 0;
        // K = G[W,W] into the leading k x k block of the (zero padded) register Sinv
        {
            double gv[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) gv[t] = Gp[wrow_s[16 * hhf + t] * gs + wrow];
#pragma unroll
            for (int t = 0; t < 16; ++t) Sr[t] = (16 * hhf + t < k && pos < k) ? gv[t] : 0.0;
        }
        // Gauss-Jordan in place (SPD: no pivoting).  Before pivot pv the swept matrix satisfies S[pv][c] = -S[c][pv] for
        // the columns already done (c < pv) and S[pv][c] = S[c][pv] for the others, so the pivot ROW every lane needs
        // is the pivot COLUMN (one register of each lane of the owning half) broadcast with that sign through LDS:
        // one write + one round trip of reads per pivot, everything else is register arithmetic.
#pragma unroll 1
        for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
            for (int tj = 0; tj < 16; ++tj) {
                const int pv = 16 * hh + tj;
                if (pv < k) {
                    if (hhf == hh) {
                        pbufa[hpos] = Sr[tj];
                        pbufb[hpos] = (hpos < pv) ? -Sr[tj] : Sr[tj];
                    }
                    wave_fence_lds();
                    const double fcol = pbufa[hpos], piv = pbufa[pv];
                    double rowpv[16];
                    bcast16(pbufb, rowpv);
                    const double ip = fast_rcp_d(piv);
                    const bool isp = hpos == pv;
                    const double f = isp ? 0.0 : fcol * ip;   // the pivot row itself is rescaled, not eliminated
                    const double scale = isp ? ip : 1.0;
#pragma unroll
                    for (int t = 0; t < 16; ++t) Sr[t] = __builtin_fma(-f, rowpv[t], Sr[t]) * scale;
                    Sr[tj] = (hhf == hh) ? (isp ? ip : -f) : Sr[tj];  // column pv
                }
            }
        }
        recompute(m);
        if (pass == 1 || k0 > 32) break;
        {
            const bool mine = lane < 32 && pos < k;
            const double viol = mine ? ((wsd > 0) ? -lam : lam) : -__builtin_inf();
            const double lmax = wave_max(mine ? fabs(lam) : 0.0);
            const bool bad = mine && viol > 1e-12 * fmax(1.0, lmax);
            const unsigned badm = (unsigned)__ballot(bad);
            const int nb = __popc(badm);
            if (nb < 2) break;
            const unsigned usedm = (k >= 32) ? 0xffffffffu : ((1u << k) - 1u);
            const unsigned keepm = usedm & ~badm;
            // rows that leave the set: clear their row-distributed flags (marker vector through rowbuf)
            put_rows(0.0, 0.0);
            if (bad) rowbuf[wrow] = 1.0;
            wave_fence_lds();
            {
                const d2 mk = *reinterpret_cast<const d2*>(rowbuf + r0);
                if (mk[0] != 0.0) act0 = false;
                if (mk[1] != 0.0) act1 = false;
            }
            // the others move up to consecutive positions
            const bool keep = lane < 32 && ((keepm >> pos) & 1u) != 0u;
            const int np = __popc(keepm & ((1u << pos) - 1u));
            if (keep) { ibuf[np] = wrow; ibuf[64 + np] = wsd; pbufa[np] = wbnd; }
            wave_fence_lds();
            k = __popc(keepm);
            wrow = 0; wsd = 0; wbnd = 0.0; lam = 0.0;
            if (pos < k) { wrow = ibuf[pos]; wsd = ibuf[64 + pos]; wbnd = pbufa[pos]; }
            wave_fence_lds();
            it += nb;
        }
        }  // pass
        if (k0 > 32) {
            tier2_home(k0);
            to_global();
            for (int q = 32; q < k0; ++q)
                add_row(PolishMode<true>{}, __builtin_amdgcn_readlane(pend_row, q), readlane_d(pend_bnd, q),
                        __builtin_amdgcn_readlane(pend_sd, q));
            fresh = false;
            overflow = true;  // continue in global mode below
        }
        }  // !started
    }
    ALMPC_STAMP(inst, 10);
    if (!give_up) {
        if (!overflow) {
            run(PolishMode<false>{});
            if (overflow) { tier2_home(33); to_global(); }
        }
        if (overflow) {
            overflow = false;
            run(PolishMode<true>{});
            if (overflow && own_slot) {  // outgrew the shared slot: continue in the global scratch
                slot_to_global();
                overflow = false;
                run(PolishMode<true>{});
            }
            give_up = overflow;  // outgrew 64
        }
    }
    if constexpr (GLDS) {
        if (own_slot) {  // done with Sinv: hand the shared slot back
            if (lane == 0) atomicExch(qcnt + 1, -1);
            own_slot = false;
        }
    }

    ALMPC_STAMP(inst, 11);
    ALMPC_ACC_FLUSH(inst);
    d2 wout;
    if (p.unsolved && lane == 0 && ((give_up || fin != 0) ? st_in : 0) != 0) {   // (rare: the redo is then launched at the next host sync)
        __hip_atomic_fetch_add(p.unsolved, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (p.redo_gate) *GL(p.redo_gate) = p.step_serial;   // (every writer of a step stores the same value)
    }
    if (give_up) {  // keep the (feasible) ADMM iterate; status stays what ADMM reported
        wout[0] = skip ? z0 : fmin(fmax(z0, lo0), hi0);
        wout[1] = skip ? z1 : fmin(fmax(z1, lo1), hi1);
    } else {
        wout[0] = act0 ? bnd0 : fmin(fmax(w0, lo0), hi0);
        wout[1] = act1 ? bnd1 : fmin(fmax(w1, lo1), hi1);
        if (lane == 0) {
            GL(p.piters)[inst] = it;
            GL(p.status)[inst] = (fin == 0) ? 0 : st_in;
        }
    }
    if (p.dflag && lane == 0 && GL(p.dflag)[inst] != 0) GL(p.status)[inst] = 2;   // (after the finish's own status write, same lane)
    if (!p.fuse_rollout) {
        if (inrow) *reinterpret_cast<d2*>(GL(p.w) + base + r0) = wout;
        return;
    }
    // ---- fused rollout: outputs of calculate! (src/main/computation_mpc.jl:50-53) for this instance
    {
        const RolloutParams& rp = p.roll;
        const int n = rp.n, m = rp.m, N = rp.N, C = n + m;
        double* Z = wave_lds;  // (N+1) x C trajectory buffer over the (now dead) active-set buffers; sized by the host
        const d2 dvp = *reinterpret_cast<const d2*>(cd + rc);
        // references first, on separate code paths for their two homes (LDS copy of a shared reference / global memory): a
        // select of the two addresses becomes a flat load that waits for every store in flight before it
        double ur0 = 0.0, ur1 = 0.0;
        if (uref_sh) {
            ur0 = shc[SL.off_uref + (in0 ? r0 : 0)];
            ur1 = shc[SL.off_uref + (in1 ? r1 : 0)];
            asm volatile("" : "+v"(ur0), "+v"(ur1));  // keeps the two paths apart (see the prologue)
        } else {
            const double* ug = GL(rp.uref) + (size_t)inst * rp.uref_stride;
            ur0 = ug[in0 ? r0 : 0];
            ur1 = ug[in1 ? r1 : 0];
        }
        if (p.fuse_rollout == 3) {
            const bool exact = p.roll_s * m <= 20 && n <= 12;   // the benchmark shape's exact fit
            double cu[ROLL_SMX], cx[ROLL_NX];
            // coefficients first: their L2 round trip runs under the u / e_u code below
            if (exact) {
                double cu2[20], cx2[12];
                roll_load<20, 12>(GL(p.rollM), lane, cu2, cx2);
#pragma unroll
                for (int c = 0; c < ROLL_SMX; ++c) cu[c] = c < 20 ? cu2[c] : 0.0;
#pragma unroll
                for (int c = 0; c < ROLL_NX; ++c) cx[c] = c < 12 ? cx2[c] : 0.0;
            } else {
                roll_load<ROLL_SMX, ROLL_NX>(GL(p.rollM), lane, cu, cx);
            }
            // blocked rollout: e_u goes to LDS in stage order (= row order), zero padded to whole blocks
            const int m0 = r0 % m, m1 = (m0 + 1 == m) ? 0 : m0 + 1;
            double e0 = 0.0, e1 = 0.0;
            if (in0) {
                const double uu = fmin(fmax(wout[0] * dvp[0] + ur0, shc[SL.off_umin + m0]), shc[SL.off_umax + m0]);
                e0 = uu - ur0;
                GL(rp.u)[(size_t)inst * nz + r0] = uu;
                GL(rp.eu)[(size_t)inst * nz + r0] = e0;
            }
            if (in1) {
                const double uu = fmin(fmax(wout[1] * dvp[1] + ur1, shc[SL.off_umin + m1]), shc[SL.off_umax + m1]);
                e1 = uu - ur1;
                GL(rp.u)[(size_t)inst * nz + r1] = uu;
                GL(rp.eu)[(size_t)inst * nz + r1] = e1;
            }
            double* eub = wave_lds;  // 256 doubles: rowbuf | pbufa | pbufb (the active-set state is dead)
            {
                d2 ev; ev[0] = e0; ev[1] = e1;
                *reinterpret_cast<d2*>(eub + 2 * lane) = ev;
                d2 zz; zz[0] = 0.0; zz[1] = 0.0;
                *reinterpret_cast<d2*>(eub + 128 + 2 * lane) = zz;
            }
            double e0v = 0.0;
            const double* xg = nullptr;
            if (lane < n) {
                double xr0;
                if (xref_sh) { xr0 = shc[SL.off_xref + lane]; asm volatile("" : "+v"(xr0)); }
                else xr0 = GL(rp.xref)[(size_t)inst * rp.xref_stride + lane];
                e0v = x0r - xr0;
            }
            if (!xref_sh) xg = GL(rp.xref) + (size_t)inst * rp.xref_stride;
            wave_fence_lds();
            ALMPC_STAMP(inst, 12);
            const size_t xo3 = (size_t)inst * n * (N + 1);
            double* gx = GL(rp.x) + xo3;
            double* gex = GL(rp.ex) + xo3;
            const double* xl = xref_sh ? (shc + SL.off_xref) : nullptr;
            if (exact) {
                double cu2[20], cx2[12];
#pragma unroll
                for (int c = 0; c < 20; ++c) cu2[c] = cu[c];
#pragma unroll
                for (int c = 0; c < 12; ++c) cx2[c] = cx[c];
                roll_run<20, 12>(cu2, cx2, n, m, N, p.roll_s, p.roll_nb, lane, eub, e0v, x0r, xl, xg, gx, gex);
            } else {
                roll_run<ROLL_SMX, ROLL_NX>(cu, cx, n, m, N, p.roll_s, p.roll_nb, lane, eub, e0v, x0r, xl, xg, gx, gex);
            }
            ALMPC_STAMP(inst, 13);
            ALMPC_STAMP(inst, 14);
            return;
        }
        const int m0 = r0 % m, m1 = (m0 + 1 == m) ? 0 : m0 + 1, k0s = r0 / m, k1s = (m0 + 1 == m) ? k0s + 1 : k0s;
        if (in0) {
            const double uu = fmin(fmax(wout[0] * dvp[0] + ur0, shc[SL.off_umin + m0]), shc[SL.off_umax + m0]);
            GL(rp.u)[(size_t)inst * nz + r0] = uu;
            GL(rp.eu)[(size_t)inst * nz + r0] = uu - ur0;
            Z[(size_t)k0s * C + n + m0] = uu - ur0;
        }
        if (in1) {
            const double uu = fmin(fmax(wout[1] * dvp[1] + ur1, shc[SL.off_umin + m1]), shc[SL.off_umax + m1]);
            GL(rp.u)[(size_t)inst * nz + r1] = uu;
            GL(rp.eu)[(size_t)inst * nz + r1] = uu - ur1;
            Z[(size_t)k1s * C + n + m1] = uu - ur1;
        }
        if (p.fuse_rollout == 2) return;  // time-varying models: inputs only (there is no single (A, B) to roll out; x is the caller's)
        if (lane < n) {
            double xr0;
            if (xref_sh) { xr0 = shc[SL.off_xref + lane]; asm volatile("" : "+v"(xr0)); }
            else xr0 = GL(rp.xref)[(size_t)inst * rp.xref_stride + lane];
            Z[lane] = x0r - xr0;
        }
        wave_fence_lds();
        ALMPC_STAMP(inst, 12);
        switch (p.roll_cpl) {
            case 1: rollout_steps<1>(Z, n, m, N, p.roll_g, lane, cab, cab + n * n); break;
            case 2: rollout_steps<2>(Z, n, m, N, p.roll_g, lane, cab, cab + n * n); break;
            case 4: rollout_steps<4>(Z, n, m, N, p.roll_g, lane, cab, cab + n * n); break;
            default: rollout_steps<8>(Z, n, m, N, p.roll_g, lane, cab, cab + n * n); break;
        }
        ALMPC_STAMP(inst, 13);
        const int nx = n * (N + 1);
        const size_t xo = (size_t)inst * nx;
        // all values first (LDS trajectory, references from their home -- two code paths, see above), then the stores back to back
        const float inv_n = 1.0f / (float)n;   // t / n for t < 4096, n <= 64: (t + 0.5) / n is never within 1e-3 of an integer
        for (int t0 = 0; t0 < nx; t0 += 4 * 64) {
            double ev[4], xr[4];
            int tt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + 64 * u + lane;
                tt[u] = t < nx ? t : nx - 1;
                const int kq = (int)(((float)tt[u] + 0.5f) * inv_n);
                ev[u] = Z[(size_t)tt[u] + (size_t)kq * m];   // (t / n) * C + t % n
            }
            if (xref_sh) {
#pragma unroll
                for (int u = 0; u < 4; ++u) xr[u] = shc[SL.off_xref + tt[u]];
                asm volatile("" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]));
            } else {
                const double* xg = GL(rp.xref) + (size_t)inst * rp.xref_stride;
#pragma unroll
                for (int u = 0; u < 4; ++u) xr[u] = xg[tt[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + 64 * u + lane;
                if (t < nx) {
                    GL(rp.ex)[xo + t] = ev[u];
                    GL(rp.x)[xo + t] = (t < n) ? x0r : ev[u] + xr[u];  // t < n <= 64 happens in the first pass only, where t == lane
                }
            }
        }
        ALMPC_STAMP(inst, 14);
    }
  };  // process

    if constexpr (GLDS) {
        // work queue: slots are handed out in the order k_admm wrote them (hard instances first); every wave leaves once
        // the counter has passed the batch, so the grid always drains
        ALMPC_STAMP(8192 + blockIdx.x * 8 + wv, 1);
        [[maybe_unused]] int nproc = 0;  // read by the diagnostic (ALMPC_STAMPS) build only
        // work queue of this workgroup: it owns the ADMM tiles blockIdx.x, blockIdx.x + gridDim.x, ... (workgroup b of
        // k_admm and of this kernel land on the same XCD, so the iterates it reads are still in that L2) and walks their
        // instances in the per-tile order k_admm left (hardest first); wave w takes i = w, afterwards the waves pull
        // the next i from an LDS counter.  No global atomics: 2048 waves popping one device-scope counter cost more
        // than the polish itself.  Every wave leaves once i runs past the last tile, so the grid always drains.
        for (int i = wv;;) {
            const int tile = (i >> 4) * (int)gridDim.x + (int)blockIdx.x;
            if (tile >= p.ntiles) break;
            const int inst = p.perm[tile * 16 + (i & 15)];
            if (inst >= 0) {
                process(inst);
                ++nproc;
            }
            if (lane_k == 0) i = atomicAdd(qcnt, 1);
            i = __builtin_amdgcn_readfirstlane(i);
        }
        ALMPC_STAMP(8192 + blockIdx.x * 8 + wv, 2);
#ifdef ALMPC_STAMPS
        if (g_stamps && (threadIdx.x & 63) == 0) g_stamps[(size_t)(8192 + blockIdx.x * 8 + wv) * 16 + 3] = nproc;
#endif
    } else {
        // slot s -> (tile s % ntiles, rank s / ntiles): the dispatch order starts with the hardest instance of every tile
        const int slot = blockIdx.x * NWV + wv;
        if (p.direct) {
            if (slot < p.batch) process(slot);
        } else if (slot < p.ntiles * 16) {
            const int inst = p.perm[(slot % p.ntiles) * 16 + slot / p.ntiles];
            if (inst >= 0) process(inst);
        }
    }
}

template <bool GLDS>
__global__ __launch_bounds__(64 * (GLDS ? POLISH_WAVES_GLDS : POLISH_WAVES))
__attribute__((amdgpu_waves_per_eu(GLDS ? POLISH_WAVES_GLDS / 4 : 2, GLDS ? POLISH_WAVES_GLDS / 4 : 2)))
void k_polish(PolishParams p_arg) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    polish_body<GLDS, false, 0>(p_arg, smem);
}

// per-instance models, small batches: single-wave workgroups, G_i and the second-tier Sinv in LDS (one workgroup per CU)
// (a template only so that it is compiled where it is instantiated -- almpc_tu_step.hip -- and not by every file that reads this header)
// START = 1: the build that can install a guessed working set whose inverse was made ahead of it (PolishParams::start_rows)
template <int START = 0>
__global__ __launch_bounds__(64) void k_polish_sgl(PolishParams p_arg) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    polish_body<false, false, 0, true, START != 0>(p_arg, smem);
}

// ------------------------------------------------------------------------------------------------
// One kernel per step for shapes whose tile is 8 waves (nz in 113..128): workgroup b first runs the ADMM of tile b, then
// polishes the 16 instances of that tile with the tile-local queue of k_polish<true> -- the two kernels already pair up
// workgroup b with tile b.  What the fusion buys: one launch instead of two, and the copy of G into LDS is requested
// under the ADMM phase (right after its prologue's own loads) with direct global -> LDS loads (global_load_lds_dwordx4: no
// registers, 1 KB per wave instruction), so it is there when the polish starts.  LDS: [G | union(ADMM buffers, polish buffers)].
// ------------------------------------------------------------------------------------------------
constexpr int STEP_KOFF = (int)((sizeof(AdmmParams) + 7) & ~size_t(7));  // PolishParams follows AdmmParams in the kernarg segment

template <int NRB, int KS>
__global__ __launch_bounds__(64 * NRB) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_step_fused(AdmmParams ap, PolishParams pp) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if constexpr (NRB != POLISH_WAVES_GLDS) return;   // (experimental builds with another polish width have no fused step)
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // (wave-uniform, and the compiler knows it)
    const int gs = (pp.nz + 1) & ~1, hs = gs / 2;  // rows of G packed to stride gs in LDS (as polish_body<true, ...> reads them)
    auto request_g = [&]() __attribute__((always_inline)) {
        for (int r = wv; r < pp.nz; r += NRB) {
            if (lane < hs) {  // one row per wave instruction: hs lanes x 16 bytes
                const char* src = reinterpret_cast<const char*>(pp.G + (size_t)r * pp.nzs) + lane * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(smem + (size_t)r * gs), 16, 0, 0);
            }
        }
    };
    admm_body<NRB, KS>(ap, smem + (size_t)pp.nz * gs, request_g);
    __builtin_amdgcn_s_waitcnt(0);  // this wave's pieces of G have landed
    __syncthreads();                // ... and everybody's; the ADMM results of the tile are visible to the whole workgroup
    polish_body<true, true, STEP_KOFF>(pp, smem);
}

// ------------------------------------------------------------------------------------------------
// rollout: outputs of calculate! (src/main/computation_mpc.jl:50-53)
//   e_u = reshape(d .* w), u = e_u + u_ref, e_x[:,1] = x0 - x_ref[:,1], e_x[:,k+1] = A e_x[:,k] + B e_u[:,k],
//   x = e_x + x_ref.   One wave per instance, lane i < n owns state row i.
// ------------------------------------------------------------------------------------------------

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_rollout(RolloutParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n = p.n, m = p.m, N = p.N;
    const int nz = m * N, nx = n * (N + 1);
    double* As = smem;                 // [n][n]  As[j*n + i] = A[i][j] (column-major as given)
    double* Bs = As + n * n;           // [m][n]
    double* wbuf = Bs + n * m;         // per wave: e_x trajectory (nx) + e_u (nz)
    ALMPC_STAMP(blockIdx.x * WAVES + (threadIdx.x >> 6), 4);
    // per-instance models: one instance per workgroup (WAVES = 1), so the workgroup's model is instance blockIdx.x's
    const double* Ag = p.A + (size_t)blockIdx.x * WAVES * p.A_stride;
    const double* Bg = p.B + (size_t)blockIdx.x * WAVES * p.B_stride;
    for (int t = threadIdx.x; t < n * n; t += blockDim.x) As[t] = Ag[t];
    for (int t = threadIdx.x; t < n * m; t += blockDim.x) Bs[t] = Bg[t];
    __syncthreads();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int inst = blockIdx.x * WAVES + wv;
    if (inst >= p.batch) return;
    ALMPC_STAMP(inst, 0);
    double* e = wbuf + (size_t)wv * (nx + nz);
    double* v = e + nx;
    // inputs: u, e_u (contiguous per instance: coalesced)
    for (int r = lane; r < nz; r += 64) {
        const double ur = p.uref[(size_t)inst * p.uref_stride + r];
        const double uu = fmin(fmax(p.w[(size_t)inst * p.nzs + r] * p.dvec[(size_t)inst * p.d_stride + r] + ur, p.umin[r % m]), p.umax[r % m]);
        const double ev = uu - ur;
        v[r] = ev;
        p.eu[(size_t)inst * nz + r] = ev;
        p.u[(size_t)inst * nz + r] = uu;
    }
    if (lane < n) e[lane] = p.x0[(size_t)inst * n + lane] - p.xref[(size_t)inst * p.xref_stride + lane];
    wave_fence_lds();
    ALMPC_STAMP(inst, 1);
    // recursion e+ = A e + B v_k into the LDS trajectory; coefficient and vector reads go out in groups of four so
    // that their LDS round trips overlap (out-of-range columns: index clamped, coefficient forced to zero)
    const int li = lane < n ? lane : 0;
    for (int k = 0; k < N; ++k) {
        const double* ek = e + (size_t)k * n;
        const double* vk = v + (size_t)k * m;
        double acc = 0.0;
        for (int j0 = 0; j0 < n; j0 += 4) {
            double a[4], x[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = (j0 + t < n) ? j0 + t : n - 1;
                a[t] = As[j * n + li];
                x[t] = ek[j];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc += ((j0 + t < n) ? a[t] : 0.0) * x[t];
        }
        for (int j0 = 0; j0 < m; j0 += 4) {
            double a[4], x[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = (j0 + t < m) ? j0 + t : m - 1;
                a[t] = Bs[j * n + li];
                x[t] = vk[j];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc += ((j0 + t < m) ? a[t] : 0.0) * x[t];
        }
        if (lane < n) e[(size_t)(k + 1) * n + lane] = acc;
        wave_fence_lds();
    }
    ALMPC_STAMP(inst, 2);
    // outputs: e_x and x = e_x + x_ref, contiguous per instance (x[:,1] is x0 itself)
    const size_t xo = (size_t)inst * nx;
    for (int t = lane; t < nx; t += 64) {
        const double ev = e[t];
        p.ex[xo + t] = ev;
        p.x[xo + t] = (t < n) ? p.x0[(size_t)inst * n + t] : ev + p.xref[(size_t)inst * p.xref_stride + t];
    }
    ALMPC_STAMP(inst, 3);
}

// Closed loop on the device: x0 <- A x0 + B u[:,1] for every instance (the plant the controller was designed for), so that a
// receding-horizon run needs no host round trip between steps.  One thread per (instance, state).
inline __global__ __launch_bounds__(256) void k_advance_plant(int n, int m, int N, int batch, const double* A, const double* B,
                                                      const double* u, const double* x0_in, double* x0) {
    extern __shared__ __attribute__((aligned(16))) double smem[];  // old x0 of the block's instances
    const int per_block = blockDim.x / n;  // instances per block
    const int li = threadIdx.x / n, i = threadIdx.x % n;
    const int inst = blockIdx.x * per_block + li;
    const bool ok = li < per_block && inst < batch;
    if (ok) smem[li * n + i] = x0_in[(size_t)inst * n + i];
    __syncthreads();
    if (!ok) return;
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += A[(size_t)j * n + i] * smem[li * n + j];
    for (int a = 0; a < m; ++a) s += B[(size_t)a * n + i] * u[(size_t)inst * m * N + a];
    x0[(size_t)inst * n + i] = s;
}

// Test hook: fill the LDS of every CU with NaN bit patterns so that a kernel that reads LDS it did not write shows up
// as a wrong result instead of passing on stale finite values (tests/test_gpu_parity.py poisons before it solves).
inline __global__ __launch_bounds__(1024) void k_poison_lds(unsigned long long pattern, int words, unsigned long long* sink) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    unsigned long long* w = reinterpret_cast<unsigned long long*>(smem);
    for (int t = threadIdx.x; t < words; t += blockDim.x) w[t] = pattern;
    __syncthreads();
    if (threadIdx.x == 0 && w[words - 1] != pattern) sink[0] = w[0];  // keep the stores alive
}

}  // namespace almpc
