// almpc_instance.hip.h -- per-instance models (almpc_design_batched): every instance has its own (A_i, B_i) and therefore its
// own condensed Hessian, scaling, KKT inverse and H'^-1.  This is the regime of BASELINE.json configs[3] (a black-box model
// re-linearised at every instance's current state): the reference's QP for given (A, B)
// (src/sub/model_modeler_implementation/linear/mpc_modeler_implementation_linear.jl:48-100, src/sub/design_mpc.jl:405-468) is
// the same, only nothing is shared across the batch any more, so there is no 16-instance MFMA tile with a common A operand.
//
//   design : the kernels of almpc_design.hip.h with blockIdx.y = instance (MFMA contraction Gamma' Qbar Gamma per instance)
//   k_admm_inst : ONE WORKGROUP PER INSTANCE AT A TIME (persistent workgroups).  The instance's KKT inverse M_i^-1 (nz x nzs
//            doubles, 115 KB for nz = 120) is streamed from HBM once, straight into the registers of the workgroup's 512
//            threads, and all K iterations run from there: HBM traffic per instance-step is one read of M_i^-1
//            (+ F'_i, V_i) instead of K reads, and the stream of the next instance overlaps the iterations of the
//            current one (second register set).
//   polish / rollout : k_polish<false> with per-instance strides for G, d, A, B (almpc_kernels.hip.h).
// Iteration formulas, termination test and outputs are those of k_admm (OSQP Algorithm 1, box form).
#pragma once
#include "almpc_kernels.hip.h"
#include "almpc_design.hip.h"   // design_scale_body
#include "almpc_fnn.hip.h"      // FnnParams

namespace almpc {

struct AdmmInstParams {
    int nz, n, m, batch, nzs;
    const double* Minv;   // [batch][nz][nzs]  (H'_i + sigma I + diag(rho_i))^-1, symmetric -- or, for k_admm_inst<true>, its packed
                          // lower triangle [batch][minv_stride], column by column (packed_tri_off): HALF the bytes of the step's stream
    long minv_stride = 0; // doubles per instance of the packed form (even)
    // admm_wave_body (k_step_inst_wave) only: doubles between two instances' operands -- nz nzs / n nzs / nzs / nz for models per
    // instance, ZERO for a model shared by the batch (round 5: the one-kernel step of small shared problems)
    long mat_stride = 0, fv_stride = 0, vec_stride = 0, fs_stride = 0;
    const double* Hs;     // [batch][nz][nzs]  H'_i (warm start only)
    const double* Fs;     // [batch][n][nzs]   F'_i = D_i F_i, column-major
    const double* Vs;     // [batch][n][nzs]   V_i = -H'_i^-1 F'_i
    const double* dvec;   // [batch][nzs]
    const double* rhovec; // [batch][nzs]
    const double* fS;     // [batch][nz]  scaled constant part of the gradient
    const double* v0S;    // [batch][nz]  -H'_i^-1 fS_i
    const double* umin; const double* umax;
    const double* uref; long uref_stride;
    const double* xref; long xref_stride;
    const double* x0;
    double* xs; double* zs; double* ys; double* v0;
    int32_t* status; int32_t* iters; int32_t* piters; int32_t* perm;
    double sigma, alpha, eps_abs, eps_rel;
    int max_iter, check_every, warm;
};

#ifdef ALMPC_STAMPS
#define ISTAMP(K) do { const int ord_ = (inst - (int)blockIdx.x) / (int)gridDim.x; if (ord_ >= 1 && ord_ < 3 && wv == 0) ALMPC_STAMP(blockIdx.x, (ord_ - 1) * 8 + (K)); } while (0)
#else
#define ISTAMP(K) do { } while (0)
#endif
constexpr int ADMM_INST_THREADS = 256;  // 4 waves, one per SIMD
constexpr int ADMM_INST_PPW = 16;       // column PAIRS per wave, compile-time bound: nz <= 128 -> 64 pairs over 4 waves

// LDS of k_admm_inst (doubles): [packed triangle (PACKED only)] | 3 x 4 partial vectors | e0 | bounds
inline size_t admm_inst_lds_doubles(int nz, int nzs, int m, bool packed) {
    return (packed ? (size_t)packed_tri_doubles(nz) + 2 : 0) + 12 * (size_t)nzs + 64 + 2 * (size_t)m;
}

// Persistent workgroups, TWO per CU: workgroup b solves instances b, b + gridDim.x, ...
// The KKT inverse of an instance never touches LDS.  Lane l owns the row pair (2l, 2l+1) and wave w the column pairs
// (2j, 2j+1), j = w, w + 4, ...: the 30 x 16 bytes a thread needs are exactly what coalesced 16-byte-per-lane loads deliver
// (1 KB per wave instruction), so the matrix streams HBM -> registers and stays there for all K iterations.  Issuing that
// stream blocks the issuing waves for about as long as HBM takes to deliver it (the vector-memory queues fill up), so a
// workgroup cannot overlap its own stream with its own iterations; the SECOND workgroup on the CU does: while one streams,
// the other iterates.  Every wave carries the whole iterate redundantly (two rows per lane), so the right-hand side
// entries of column pair j are two v_readlane pairs from lane j; only the 4 partial vectors of a product cross waves
// (LDS, double buffered: ONE barrier per iteration).
//
// PACKED (round 5): M_i^-1 is symmetric, and streaming it whole reads every off-diagonal element twice.  Here the instance's matrix
// is its packed lower triangle (58 KB instead of 123 KB for nz = 120).  Lane l needs elements (2l, c) and (2l + 1, c) for its wave's
// columns c -- for c above the rows that is a strided walk over the triangle, which no coalesced load delivers -- so the triangle
// lands in LDS first (direct global -> LDS loads, 1 KB per wave instruction, no registers) and every lane gathers its 64 elements
// from there ONCE per instance-step (ds_read_b64 at (max, min)); the iterations run from registers exactly as before.  The stream of
// the NEXT instance is requested as soon as the gather is done and lands under this instance's iterations (the LDS-only barriers of
// the loop do not wait for it); every other global load of an instance is therefore issued BEFORE that request (the memory counter
// retires in order: a later load could not be waited for without waiting for the 58 KB as well).
template <bool PACKED>
__global__ __launch_bounds__(ADMM_INST_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_admm_inst(AdmmInstParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem_all[];
    const long tri_doubles = PACKED ? packed_tri_doubles(p.nz) : 0;
    double* tri = smem_all;                 // [tri_doubles | 0 0] packed triangle of the instance being set up, two zeros behind it
    double* smem = smem_all + tri_doubles + (PACKED ? 2 : 0);
    if (PACKED && threadIdx.x == 0) { tri[tri_doubles] = 0.0; tri[tri_doubles + 1] = 0.0; }   // (visible after the first barrier)
    const int nz = p.nz, nzs = p.nzs, n = p.n;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, and the compiler knows it: column offsets stay scalar
    constexpr int NW = ADMM_INST_THREADS / 64;  // 4 waves
    constexpr int PPW = ADMM_INST_PPW;
    constexpr int PFC = 4;                  // columns of F'_i / V_i per wave loaded with the matrix (n <= 16; more: separate loop)
    double* part0 = smem;                   // [4][nzs] partial products of the waves (two buffers)
    double* part1 = part0 + NW * nzs;
    double* partv = part1 + NW * nzs;       // [4][nzs] V e0 partials (beside F' e0 in part0) at the start of an instance
    double* e0s = partv + NW * nzs;         // [n <= 64]
    double* bnd = e0s + 64;                 // [2 m] umin | umax
    const int r0 = 2 * lane, r1 = 2 * lane + 1;
    const bool inpair = r0 < nzs;           // nzs is even: both rows of the pair are inside the padded vector or neither
    const int rc = inpair ? r0 : 0;
    const bool own0 = r0 < nz, own1 = r1 < nz;
    const int npairs = (nz + 1) / 2;

    for (int t = tid; t < p.m; t += ADMM_INST_THREADS) { bnd[t] = p.umin[t]; bnd[p.m + t] = p.umax[t]; }
    // PACKED: request the triangle of instance i into LDS: wave w takes the 1 KB pieces w, w + 4, ...
    auto request_tri = [&](const double* minv, long stride, int i) __attribute__((always_inline)) {
        const char* src = reinterpret_cast<const char*>(minv + (size_t)i * stride);
        const int bytes = (int)(tri_doubles * 8);
        // Inline assembly on purpose: the compiler treats a global -> LDS load it knows about as a pending LDS write and makes every
        // later LDS barrier and LDS read wait for it (vmcnt(0)) -- the stream would not run under the iterations at all.  It does not
        // look inside the asm; the one wait this stream needs is the explicit s_waitcnt in front of the gather.  M0 carries the LDS
        // address (the hardware adds 16 bytes per lane); the s_nop is the wait state between a scalar write of M0 and the instruction.
        const uint32_t tri_lds = (uint32_t)(uintptr_t)tri;
        for (int off = wv * 1024; off < bytes; off += NW * 1024)
            if (off + lane * 16 < bytes)
                asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src + off + lane * 16), "{m0}"(__builtin_amdgcn_readfirstlane(tri_lds + (uint32_t)off)) : "memory");
    };
    if constexpr (PACKED) {
        if ((int)blockIdx.x < p.batch) request_tri(GL(p.Minv), p.minv_stride, blockIdx.x);
    }

    for (int inst = blockIdx.x; inst < p.batch; inst += gridDim.x) {
        // parameters re-read through an opaque kernarg pointer per instance: otherwise every instance-independent load is
        // hoisted out of the persistent loop and the kernel spills (same measure as in k_polish)
        auto ka_ = __builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka_));
        const AdmmInstParams& q = *(const AdmmInstParams*)ka_;
        ISTAMP(0);
        // ---- one batch of loads: this wave's column pairs of M_i^-1, its columns of F'_i and V_i, the row constants, e0
        d2 mcol[2 * PPW];
        if constexpr (!PACKED) {
            const double* Mi = GL(q.Minv) + (size_t)inst * nz * nzs;
#pragma unroll
            for (int u = 0; u < PPW; ++u) {
                const int j = wv + NW * u;                  // column pair (2j, 2j+1)
                const int c0 = 2 * j < nz ? 2 * j : 0, c1 = 2 * j + 1 < nz ? 2 * j + 1 : 0;
                mcol[2 * u] = *reinterpret_cast<const d2*>(Mi + (size_t)c0 * nzs + rc);
                mcol[2 * u + 1] = *reinterpret_cast<const d2*>(Mi + (size_t)c1 * nzs + rc);
            }
        }
        const int ra = own0 ? r0 : 0, rb = own1 ? r1 : 0;
        d2 cf[PFC], cv[PFC], pdd, prh;
        double ur0, ur1, fS0, fS1, vS0, vS1, e0v;
        auto issue_small = [&](int ii) __attribute__((always_inline)) {   // this wave's columns of F'_i and V_i, the row constants, e0
            const double* Fi = GL(q.Fs) + (size_t)ii * n * nzs;
            const double* Vi = GL(q.Vs) + (size_t)ii * n * nzs;
#pragma unroll
            for (int u = 0; u < PFC; ++u) {
                const int c = wv + NW * u;
                const int cc = c < n ? c : 0;
                cf[u] = *reinterpret_cast<const d2*>(Fi + (size_t)cc * nzs + rc);
                cv[u] = *reinterpret_cast<const d2*>(Vi + (size_t)cc * nzs + rc);
            }
            pdd = *reinterpret_cast<const d2*>(GL(q.dvec) + (size_t)ii * nzs + rc);
            prh = *reinterpret_cast<const d2*>(GL(q.rhovec) + (size_t)ii * nzs + rc);
            ur0 = GL(q.uref)[(size_t)ii * q.uref_stride + ra]; ur1 = GL(q.uref)[(size_t)ii * q.uref_stride + rb];
            fS0 = GL(q.fS)[(size_t)ii * nz + ra]; fS1 = GL(q.fS)[(size_t)ii * nz + rb];
            vS0 = GL(q.v0S)[(size_t)ii * nz + ra]; vS1 = GL(q.v0S)[(size_t)ii * nz + rb];
            e0v = 0.0;
            if (tid < n) e0v = GL(q.x0)[(size_t)ii * n + tid] - GL(q.xref)[(size_t)ii * q.xref_stride + tid];
        };
        // (measured and dropped: requesting these for the NEXT instance right behind the iterations -- live across that instance's
        // gather they are spilled, 52 registers)
        issue_small(inst);   // (full layout: with the matrix stream, one batch of loads; PACKED: they land under the wait for the triangle)
        if constexpr (PACKED) __builtin_amdgcn_s_waitcnt(0);   // this wave's pieces of the instance's triangle have landed in LDS
        lds_barrier();  // the previous instance is done with the LDS vectors (PACKED: and every wave's pieces are there)
        if (tid < n) e0s[tid] = e0v;
        if constexpr (PACKED) {
            // gather: the 2 x 2 block (rows 2l, 2l+1) x (columns 2j, 2j+1) of the symmetric matrix is the STORED block (b, a) =
            // (max(l, j), min(l, j)), two aligned 16-byte LDS reads (nz is even here; see packed_tri_off):
            //   A = column 2a at row 2b     -> (M[2b][2a],   M[2b+1][2a])
            //   B = column 2a+1 at row 2b-1 -> (M[2b][2a+1], M[2b+1][2a+1])      (b == a: the first entry is the pad behind column 2a,
            //                                                                      where the inverse kernel keeps a copy of M[2a+1][2a])
            // and, by symmetry, e00 = A0, e11 = B1, (e10, e01) = (A1, B0) on and below the diagonal, (B0, A1) above it: min / max
            // address arithmetic and ONE select mask per block.  Rows / columns beyond nz read the two zeros behind the triangle.
            // (The lane index is made opaque HERE: otherwise every address is formed above the barrier and spilled.)
            int lg = lane;
            asm volatile("" : "+v"(lg));
            const int zi = (int)tri_doubles;
#pragma unroll
            for (int u = 0; u < PPW; ++u) {
                const int j = wv + NW * u;                  // column pair (2j, 2j+1)
                const bool valid = own0 && 2 * j < nz;
                const int a = lg < j ? lg : j, b = lg < j ? j : lg;
                const int ia = valid ? packed_tri_off(nz, 2 * a) + 2 * (b - a) : zi;
                const int ib = valid ? packed_tri_off(nz, 2 * a + 1) + 2 * (b - a) - 1 : zi;
                const d2 A = *reinterpret_cast<const d2*>(tri + ia);
                const d2 B = *reinterpret_cast<const d2*>(tri + ib);
                const bool above = lg < j;
                mcol[2 * u][0] = A[0]; mcol[2 * u][1] = above ? B[0] : A[1];
                mcol[2 * u + 1][0] = above ? A[1] : B[0]; mcol[2 * u + 1][1] = B[1];
                if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        double dv[2], lo[2], hi[2], fs[2], v0[2], rho[2];
        {
            const double di0 = 1.0 / (own0 ? pdd[0] : 1.0), di1 = 1.0 / (own1 ? pdd[1] : 1.0);
            dv[0] = own0 ? pdd[0] : 1.0; dv[1] = own1 ? pdd[1] : 1.0;
            rho[0] = own0 ? prh[0] : 1.0; rho[1] = own1 ? prh[1] : 1.0;
            lo[0] = own0 ? (bnd[ra % q.m] - ur0) * di0 : 0.0; hi[0] = own0 ? (bnd[q.m + ra % q.m] - ur0) * di0 : 0.0;
            lo[1] = own1 ? (bnd[rb % q.m] - ur1) * di1 : 0.0; hi[1] = own1 ? (bnd[q.m + rb % q.m] - ur1) * di1 : 0.0;
            fs[0] = own0 ? fS0 : 0.0; fs[1] = own1 ? fS1 : 0.0;   // + F' e0 below
            v0[0] = own0 ? vS0 : 0.0; v0[1] = own1 ? vS1 : 0.0;   // + V e0 below
        }
        lds_barrier();
        if constexpr (PACKED) {
            // every wave has gathered and every global load of this instance has been waited for (the s_waitcnt above): the NEXT
            // instance's triangle may come, and lands under this instance's gradient and iterations.  (A warm start reads more of
            // this instance from global memory first -- its loads would queue behind the 58 KB -- and requests after that.)
            if (!q.warm && inst + (int)gridDim.x < q.batch) request_tri(GL(q.Minv), q.minv_stride, inst + (int)gridDim.x);
        }
        ISTAMP(1);

        // ---- f' = F' e0 + fS, v0 = V e0 + v0S: wave w multiplies columns w, w + 4, ...
        {
            d2 af = {0.0, 0.0}, av = {0.0, 0.0};
#pragma unroll
            for (int u = 0; u < PFC; ++u)
                if (wv + NW * u < n) { const double e = e0s[wv + NW * u]; af += cf[u] * e; av += cv[u] * e; }
            if (n > PFC * NW) {  // wide state vectors: the remaining columns
                const double* Fi = GL(q.Fs) + (size_t)inst * n * nzs;
                const double* Vi = GL(q.Vs) + (size_t)inst * n * nzs;
                for (int c = wv + PFC * NW; c < n; c += NW) {
                    const double e = e0s[c];
                    af += *reinterpret_cast<const d2*>(Fi + (size_t)c * nzs + rc) * e;
                    av += *reinterpret_cast<const d2*>(Vi + (size_t)c * nzs + rc) * e;
                }
            }
            if (inpair) {
                *reinterpret_cast<d2*>(part0 + wv * nzs + rc) = af;
                *reinterpret_cast<d2*>(partv + wv * nzs + rc) = av;
            }
        }
        lds_barrier();
        {
            d2 af = {0.0, 0.0}, av = {0.0, 0.0};
            if (inpair) {
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    af += *reinterpret_cast<const d2*>(part0 + w * nzs + rc);
                    av += *reinterpret_cast<const d2*>(partv + w * nzs + rc);
                }
            }
            if (own0) { fs[0] += af[0]; v0[0] += av[0]; }
            if (own1) { fs[1] += af[1]; v0[1] += av[1]; }
        }
        ISTAMP(2);
        const double nf = wave_max(fmax(fabs(fs[0] / dv[0]), fabs(fs[1] / dv[1])));  // |f/d|_inf (pad rows contribute 0)

        // ---- initial iterate (yt = y / rho_r); every wave holds all of it
        double x[2] = {0, 0}, z[2] = {0, 0}, yt[2] = {0, 0}, px[2] = {0, 0}, rown[2];
        const double sigma = q.sigma, alpha = q.alpha;
        if (q.warm) {
            if (inpair) {
                const size_t o = (size_t)inst * nzs + rc;
                const d2 xx = *reinterpret_cast<const d2*>(GL(q.xs) + o), yy = *reinterpret_cast<const d2*>(GL(q.ys) + o),
                         zz = *reinterpret_cast<const d2*>(GL(q.zs) + o);
                if (own0) { x[0] = xx[0]; yt[0] = yy[0] / rho[0]; z[0] = fmin(fmax(zz[0], lo[0]), hi[0]); }
                if (own1) { x[1] = xx[1]; yt[1] = yy[1] / rho[1]; z[1] = fmin(fmax(zz[1], lo[1]), hi[1]); }
            }
            // px = H'_i x: one product with the instance's scaled Hessian (global memory, this wave's column pairs), summed
            // like the iteration's product below
            d2 a = {0.0, 0.0};
            const double* Hi = GL(q.Hs) + (size_t)inst * nz * nzs;
            for (int j = wv; j < npairs; j += NW) {
                const double s0 = readlane_d(x[0], j), s1 = readlane_d(x[1], j);
                a += *reinterpret_cast<const d2*>(Hi + (size_t)(2 * j) * nzs + rc) * s0;
                if (2 * j + 1 < nz) a += *reinterpret_cast<const d2*>(Hi + (size_t)(2 * j + 1) * nzs + rc) * s1;
            }
            lds_barrier();  // part0 was read above
            if (inpair) *reinterpret_cast<d2*>(part0 + wv * nzs + rc) = a;
            lds_barrier();
            d2 sum = {0.0, 0.0};
            if (inpair) {
#pragma unroll
                for (int w = 0; w < NW; ++w) sum += *reinterpret_cast<const d2*>(part0 + w * nzs + rc);
            }
            px[0] = own0 ? sum[0] : 0.0; px[1] = own1 ? sum[1] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) rown[j] = sigma * x[j] - fs[j] + rho[j] * (z[j] - yt[j]);  // pad rows: 0
        lds_barrier();  // the partial buffers are free again
        if constexpr (PACKED) {   // every load of this instance has been consumed and every wave has gathered: the next triangle may come
            // (explicit vmcnt(0) first: a register loaded above and first READ below the request would otherwise be waited for with
            // vmcnt(0) there -- the request is a loop, the compiler cannot count its loads -- i.e. for the whole 58 KB, at once)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            if (q.warm && inst + (int)gridDim.x < q.batch) request_tri(GL(q.Minv), q.minv_stride, inst + (int)gridDim.x);
        }
        ISTAMP(3);
        ISTAMP(4);
        bool active = true;
        int my_iters = q.max_iter, my_status = 1;
        double* pc = part0;
        double* pn = part1;
#pragma unroll 1
        for (int it = 1; it <= q.max_iter; ++it) {
            {   // wave w: column pairs j = w, w + 4, ... of M_i^-1 (registers) times (rhs[2j], rhs[2j+1]) = the two components of lane j
                d2 acc0 = {0.0, 0.0}, acc1 = {0.0, 0.0};
#pragma unroll
                for (int u = 0; u < PPW; ++u) {
                    const int j = wv + NW * u;
                    if (j < npairs) {  // wave-uniform
                        const double s0 = readlane_d(rown[0], j), s1 = readlane_d(rown[1], j);  // rhs of a pad row is 0
                        acc0 += mcol[2 * u] * s0;
                        acc1 += mcol[2 * u + 1] * s1;
                    }
                }
                if (inpair) *reinterpret_cast<d2*>(pc + wv * nzs + rc) = acc0 + acc1;
            }
            lds_barrier();
            {
                d2 xt2 = {0.0, 0.0};
                if (inpair) {
#pragma unroll
                    for (int w = 0; w < NW; ++w) xt2 += *reinterpret_cast<const d2*>(pc + w * nzs + rc);
                }
                if (active) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const double xt = xt2[j];
                        const double hxt = rown[j] - (sigma + rho[j]) * xt;  // H' xt, from the KKT identity
                        px[j] = alpha * hxt + (1.0 - alpha) * px[j];
                        x[j] = alpha * xt + (1.0 - alpha) * x[j];
                        const double w = alpha * xt + (1.0 - alpha) * z[j] + yt[j];
                        const double zn = fmin(fmax(w, lo[j]), hi[j]);
                        yt[j] = w - zn;
                        z[j] = zn;
                        const bool own = j == 0 ? own0 : own1;
                        rown[j] = own ? sigma * x[j] - fs[j] + rho[j] * (z[j] - yt[j]) : 0.0;
                    }
                }
            }
            { double* t = pc; pc = pn; pn = t; }
            const bool check = (it % q.check_every == 0) || (it == q.max_iter);
            if (check) {  // every wave holds the whole iterate: wave-local reductions, no LDS
                double t_rp = 0, t_x = 0, t_z = 0, t_rd = 0, t_hx = 0, t_y = 0, t_bad = 0;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bool own = j == 0 ? own0 : own1;
                    if (!own) continue;
                    const double yi = rho[j] * yt[j], di = 1.0 / dv[j];
                    t_rp = fmax(t_rp, fabs(dv[j] * (x[j] - z[j])));
                    t_x = fmax(t_x, fabs(dv[j] * x[j]));
                    t_z = fmax(t_z, fabs(dv[j] * z[j]));
                    t_rd = fmax(t_rd, fabs((px[j] + fs[j] + yi) * di));
                    t_hx = fmax(t_hx, fabs(px[j] * di));
                    t_y = fmax(t_y, fabs(yi * di));
                    const double s = x[j] + yi + px[j];
                    if (!(fabs(s) <= 1.79e308)) t_bad = 1.0;
                }
                const double rp = wave_max(t_rp);
                __builtin_amdgcn_sched_barrier(0);  // one DPP chain after the other: interleaved they cost ~50 registers
                const double nx = wave_max(t_x);
                __builtin_amdgcn_sched_barrier(0);
                const double nzn = wave_max(t_z);
                __builtin_amdgcn_sched_barrier(0);
                const double rd = wave_max(t_rd);
                __builtin_amdgcn_sched_barrier(0);
                const double nhx = wave_max(t_hx);
                __builtin_amdgcn_sched_barrier(0);
                const double ny = wave_max(t_y);
                __builtin_amdgcn_sched_barrier(0);
                const double bad = wave_max(t_bad);
                if (active) {
                    const bool conv = (rp <= q.eps_abs + q.eps_rel * fmax(nx, nzn)) &&
                                      (rd <= q.eps_abs + q.eps_rel * fmax(fmax(nhx, ny), nf));
                    if (bad > 0.0) { active = false; my_iters = it; my_status = 2; }
                    else if (conv) { active = false; my_iters = it; my_status = 0; }
                }
                if (!active) break;  // identical in every wave: all of them reduced the same values in the same order
            }
        }

        ISTAMP(5);
        // ---- hand-off to the polish: same arrays as k_admm (wave 0 writes)
        if (tid == 0) {
            GL(q.iters)[inst] = my_iters;
            GL(q.status)[inst] = my_status;
            GL(q.piters)[inst] = 0;
            GL(q.perm)[inst] = inst;  // no ranking across instances here: processing order = instance order
        }
        if (wv == 0 && inpair) {
            const size_t o = (size_t)inst * nzs + rc;
            d2 a, b, c, d;
            a[0] = own0 ? x[0] : 0.0; a[1] = own1 ? x[1] : 0.0;
            b[0] = own0 ? z[0] : 0.0; b[1] = own1 ? z[1] : 0.0;
            c[0] = own0 ? rho[0] * yt[0] : 0.0; c[1] = own1 ? rho[1] * yt[1] : 0.0;
            d[0] = own0 ? v0[0] : 0.0; d[1] = own1 ? v0[1] : 0.0;
            *reinterpret_cast<d2*>(GL(q.xs) + o) = a;
            *reinterpret_cast<d2*>(GL(q.zs) + o) = b;
            *reinterpret_cast<d2*>(GL(q.ys) + o) = c;
            *reinterpret_cast<d2*>(GL(q.v0) + o) = d;
        }
        ISTAMP(6);
    }
#undef ISTAMP
}

// ------------------------------------------------------------------------------------------------
// k_step_inst_wave<NZC>: one kernel per step for SMALL per-instance problems (padded size nzs <= NZC <= 64; BASELINE configs[3]: nz 40).
// k_admm_inst is built to stream a 115 KB KKT inverse per instance through four waves; at nz = 40 the matrix is 15 KB, 1024
// instances are four per CU, and the two-launch step (137 us) is latency of barriers and of a workgroup per instance.  Here ONE wave
// owns an instance for the whole step: lane r holds row r of M_i^-1 in registers (column c of the symmetric matrix is one coalesced
// load), the right-hand side is broadcast through a 64-double LDS buffer (uniform-address reads), every reduction is a DPP chain --
// no workgroup barrier anywhere --, and the same wave then runs the exact active-set finish of polish_body (single-wave build: G_i
// and the second-tier Sinv in the wave's LDS) on the iterate it has just written.  Same arithmetic as k_admm_inst per row (the sums
// run over columns in ascending order instead of four partial sums: results agree to rounding).
template <int NZC>
__device__ __forceinline__ void admm_wave_body(const AdmmInstParams& q, const int inst, double* buf /* LDS, >= 2 * 64 doubles */) {
    const int lane = threadIdx.x & 63;
    const int nz = q.nz, nzs = q.nzs, n = q.n;
    const bool own = lane < nz;
    const int r = own ? lane : 0;
    // ---- loads: row r of M_i^-1 (= column r, symmetric: element (r, c) at c * nzs + r), the row constants, e0
    double mrow[NZC];
    {
        const double* Mi = GL(q.Minv) + (size_t)inst * q.mat_stride + r;
#pragma unroll
        for (int c = 0; c < NZC; ++c) mrow[c] = (c < nz) ? Mi[(size_t)c * nzs] : 0.0;
    }
    const double dvr = own ? GL(q.dvec)[(size_t)inst * q.vec_stride + r] : 1.0;
    const double rho = own ? GL(q.rhovec)[(size_t)inst * q.vec_stride + r] : 1.0;
    const double ur = GL(q.uref)[(size_t)inst * q.uref_stride + r];
    double fs = own ? GL(q.fS)[(size_t)inst * q.fs_stride + r] : 0.0;
    double v0 = own ? GL(q.v0S)[(size_t)inst * q.fs_stride + r] : 0.0;
    const double e0l = (lane < n) ? GL(q.x0)[(size_t)inst * n + lane] - GL(q.xref)[(size_t)inst * q.xref_stride + lane] : 0.0;
    const double dinv = 1.0 / dvr;
    const double lo = own ? (q.umin[r % q.m] - ur) * dinv : 0.0, hi = own ? (q.umax[r % q.m] - ur) * dinv : 0.0;
    // ---- f' = F'_i e0 + fS, v0 = V_i e0 + v0S: column c of F'_i / V_i is one coalesced load, e0_c comes from lane c
    {
        const double* Fi = GL(q.Fs) + (size_t)inst * q.fv_stride + r;
        const double* Vi = GL(q.Vs) + (size_t)inst * q.fv_stride + r;
        for (int c0 = 0; c0 < n; c0 += 4) {
            double fc[4], vc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = (c0 + u < n) ? c0 + u : 0;
                fc[u] = Fi[(size_t)c * nzs]; vc[u] = Vi[(size_t)c * nzs];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (c0 + u < n) { const double e = readlane_d(e0l, c0 + u); fs += fc[u] * e; v0 += vc[u] * e; }
        }
        if (!own) { fs = 0.0; v0 = 0.0; }
    }
    const double nf = wave_max(fabs(fs * dinv));   // |f/d|_inf
    double x = 0.0, z = 0.0, yt = 0.0, px = 0.0;
    const double sigma = q.sigma, alpha = q.alpha;
    if (q.warm) {
        const size_t o = (size_t)inst * nzs + r;
        if (own) { x = GL(q.xs)[o]; yt = GL(q.ys)[o] / rho; z = fmin(fmax(GL(q.zs)[o], lo), hi); }
        // px = H'_i x: one product with the instance's scaled Hessian, column by column from global memory
        const double* Hi = GL(q.Hs) + (size_t)inst * q.mat_stride + r;
        double a = 0.0;
        for (int c = 0; c < nz; ++c) a += Hi[(size_t)c * nzs] * readlane_d(x, c);
        px = own ? a : 0.0;
    }
    double rown = own ? sigma * x - fs + rho * (z - yt) : 0.0;
    bool active = true;
    int my_iters = q.max_iter, my_status = 1;
    double* pc = buf;
    double* pn = buf + 64;
#pragma unroll 1
    for (int it = 1; it <= q.max_iter; ++it) {
        pc[lane] = rown;          // pad rows: 0
        wave_fence_lds();
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int c = 0; c < NZC; c += 2) {   // uniform-address reads: the right-hand side broadcast to every lane, two entries per read
            const d2 rv = *reinterpret_cast<const d2*>(pc + c);
            acc0 = __builtin_fma(mrow[c], rv[0], acc0);
            acc1 = __builtin_fma(mrow[c + 1], rv[1], acc1);
        }
        const double xt = acc0 + acc1;
        if (active) {
            const double hxt = rown - (sigma + rho) * xt;  // H' xt, from the KKT identity
            px = alpha * hxt + (1.0 - alpha) * px;
            x = alpha * xt + (1.0 - alpha) * x;
            const double w = alpha * xt + (1.0 - alpha) * z + yt;
            const double zn = fmin(fmax(w, lo), hi);
            yt = w - zn;
            z = zn;
            rown = own ? sigma * x - fs + rho * (z - yt) : 0.0;
        }
        { double* t = pc; pc = pn; pn = t; }
        const bool check = (it % q.check_every == 0) || (it == q.max_iter);
        if (check) {
            const double yi = rho * yt;
            const double rp = wave_max(own ? fabs(dvr * (x - z)) : 0.0);
            __builtin_amdgcn_sched_barrier(0);
            const double nx = wave_max(own ? fabs(dvr * x) : 0.0);
            __builtin_amdgcn_sched_barrier(0);
            const double nzn = wave_max(own ? fabs(dvr * z) : 0.0);
            __builtin_amdgcn_sched_barrier(0);
            const double rd = wave_max(own ? fabs((px + fs + yi) * dinv) : 0.0);
            __builtin_amdgcn_sched_barrier(0);
            const double nhx = wave_max(own ? fabs(px * dinv) : 0.0);
            __builtin_amdgcn_sched_barrier(0);
            const double ny = wave_max(own ? fabs(yi * dinv) : 0.0);
            __builtin_amdgcn_sched_barrier(0);
            const double sfin = x + yi + px;
            const double bad = wave_max((own && !(fabs(sfin) <= 1.79e308)) ? 1.0 : 0.0);
            if (active) {
                const bool conv = (rp <= q.eps_abs + q.eps_rel * fmax(nx, nzn)) && (rd <= q.eps_abs + q.eps_rel * fmax(fmax(nhx, ny), nf));
                if (bad > 0.0) { active = false; my_iters = it; my_status = 2; }
                else if (conv) { active = false; my_iters = it; my_status = 0; }
            }
            if (!active) break;
        }
    }
    // ---- hand-off to the finish: the arrays k_admm_inst writes
    if (lane == 0) {
        GL(q.iters)[inst] = my_iters;
        GL(q.status)[inst] = my_status;
        GL(q.piters)[inst] = 0;
        GL(q.perm)[inst] = inst;
    }
    if (lane < nzs) {
        const size_t o = (size_t)inst * nzs + lane;
        GL(q.xs)[o] = own ? x : 0.0;
        GL(q.zs)[o] = own ? z : 0.0;
        GL(q.ys)[o] = own ? rho * yt : 0.0;
        GL(q.v0)[o] = own ? v0 : 0.0;
    }
}

constexpr int STEP_INST_KOFF = (int)((sizeof(AdmmInstParams) + 7) & ~size_t(7));   // PolishParams follows AdmmInstParams in the kernarg segment

template <int NZC>
__global__ __launch_bounds__(64) void k_step_inst_wave(AdmmInstParams ip, PolishParams pp) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int inst = blockIdx.x;
    if (inst < ip.batch) {
        // the ADMM phase borrows the head of the wave's polish slot (rowbuf | pbufa: 192 doubles) for its broadcast buffers
        const PolishShared SL = polish_shared_layout(pp.roll.n, pp.m, pp.roll.N, pp.nz, pp.nzs, pp.fuse_rollout);
        admm_wave_body<NZC>(ip, inst, smem + SL.total);
    }
    // the finish reads the iterate back through global memory: make this wave's stores visible to its own later loads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    polish_body<false, false, STEP_INST_KOFF, true>(pp, smem);
}

// ------------------------------------------------------------------------------------------------
// k_guess_iterate: hand-off to the active-set finish for an SQP iteration after the first one, instead of an ADMM phase.
// The QP variable is v = u - ubar around the current iterate, so v = 0 is feasible, and the rows that are active at the
// solution of this QP are, up to a few changes, the inputs that already sit ON a bound: the previous QP put them there and the
// update clamps exactly.  Guess: z = 0, working set = rows with ubar on a bound, multiplier sign = the side (the rule the finish
// reads: y > 0 on the upper bound, y < 0 on the lower).  The finish is an exact primal active-set method whatever its start, so
// the result is the same as with the ADMM guess; what goes away is the ADMM phase AND the KKT inverse it needs -- one of the two
// n_z x n_z inverses of every iteration.  One wave per instance, same output arrays as k_admm_inst.
// ------------------------------------------------------------------------------------------------
inline __global__ __launch_bounds__(256) void k_guess_iterate(AdmmInstParams p) {
    const int nz = p.nz, nzs = p.nzs, n = p.n, lane = threadIdx.x & 63;
    const int inst = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (inst >= p.batch) return;
    const double* Vi = p.Vs + (size_t)inst * n * nzs;
    const double* dv = p.dvec + (size_t)inst * nzs;
    const size_t o = (size_t)inst * nzs;
    for (int r = lane; r < nzs; r += 64) {
        double v = 0.0, z = 0.0, y = 0.0;
        if (r < nz) {
            v = p.v0S[(size_t)inst * nz + r];
            for (int c = 0; c < n; ++c)
                v += Vi[(size_t)c * nzs + r] * (p.x0[(size_t)inst * n + c] - p.xref[(size_t)inst * p.xref_stride + c]);
            const double ur = p.uref[(size_t)inst * p.uref_stride + r], di = 1.0 / dv[r];
            const double lo = (p.umin[r % p.m] - ur) * di, hi = (p.umax[r % p.m] - ur) * di;  // as k_admm_inst forms them
            z = fmin(fmax(0.0, lo), hi);
            y = (hi <= 0.0) ? 1.0 : ((lo >= 0.0) ? -1.0 : 0.0);
        }
        p.xs[o + r] = z; p.zs[o + r] = z; p.ys[o + r] = y; p.v0[o + r] = v;
    }
    if (lane == 0) {
        p.iters[inst] = 0;
        p.status[inst] = 1;   // "not converged": the finish sets 0 when it certifies the optimum
        p.piters[inst] = 0;
        p.perm[inst] = inst;
    }
}

// k_guess_iterate_ws: k_guess_iterate for nz <= 128 with the inverse of the guessed working set made on the spot (round 5).  An SQP
// iterate has about half of its inputs on a bound (50 of 100 at the configs[4] shape), and the single-wave finish builds the inverse of
// such a set as 32 rows of Gauss-Jordan in registers + one bordering in memory mode per further row, 6 k cycles each: 108 k of the
// 190 k cycles of that launch, with 255 of the chip's 256 CUs holding one wave.  Here: ONE WORKGROUP of four waves per instance; thread r
// forms row r of the guess exactly as k_guess_iterate does; the flagged rows (y != 0), ascending, are the working set; if it has 33..64
// rows, K = G_i[W, W] is gathered (lane = position, wave w = positions 16 w .. 16 w + 15 as columns) and inverted by the symmetric
// Gauss-Jordan sweep of k_sdual_start (two pivot rows at a time through LDS, double buffered: one barrier per PAIR of pivots; K is positive
// definite: no row is left out, a non-positive pivot voids the start), and written to the finish's second-tier scratch (64 x 64,
// identity padded) with the row list beside it (PolishParams::start_rows).  The finish compares the list with its own reading of the
// guess before it installs anything.
struct GuessWsParams {
    const double* G; long G_stride;   // G_i = H_i'^-1, dense [nz][nzs]
    double* sinv;                     // [batch][64 * 64]  (PolishParams::sglobal)
    int32_t* rows;                    // [batch][65] count (0: no start) + rows
};

inline __global__ __launch_bounds__(256) void k_guess_iterate_ws(AdmmInstParams p, GuessWsParams w) {
    __shared__ __attribute__((aligned(16))) double s_prow[2][128];   // two pivot rows per step, double buffered: one barrier per step
    __shared__ int s_rows[64];
    __shared__ int s_cnt[4];
    const int nz = p.nz, nzs = p.nzs, n = p.n, tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), c0 = 16 * wv;   // (wave-uniform, and the compiler knows it: the column tests below stay scalar)
    const int inst = blockIdx.x;
    const size_t o = (size_t)inst * nzs;
    // ---- the guess, one row per thread (nzs <= 128: the host's condition)
    if (wv == 0) ALMPC_STAMP(inst, 0);
#ifdef ALMPC_STAMPS
    if (g_stamps && tid == 0) g_stamps[(size_t)inst * 16 + 6] = __builtin_amdgcn_s_memrealtime();
#endif
    double y = 0.0;
    if (tid < nzs) {
        const int r = tid;
        double v = 0.0, z = 0.0;
        if (r < nz) {
            const double* Vi = p.Vs + (size_t)inst * n * nzs;
            v = p.v0S[(size_t)inst * nz + r];
            for (int c = 0; c < n; ++c)
                v += Vi[(size_t)c * nzs + r] * (p.x0[(size_t)inst * n + c] - p.xref[(size_t)inst * p.xref_stride + c]);
            const double ur = p.uref[(size_t)inst * p.uref_stride + r], di = 1.0 / p.dvec[o + r];
            const double lo = (p.umin[r % p.m] - ur) * di, hi = (p.umax[r % p.m] - ur) * di;  // as k_admm_inst forms them
            z = fmin(fmax(0.0, lo), hi);
            y = (hi <= 0.0) ? 1.0 : ((lo >= 0.0) ? -1.0 : 0.0);
        }
        p.xs[o + r] = z; p.zs[o + r] = z; p.ys[o + r] = y; p.v0[o + r] = v;
    }
    if (tid == 0) {
        p.iters[inst] = 0;
        p.status[inst] = 1;   // "not converged": the finish sets 0 when it certifies the optimum
        p.piters[inst] = 0;
        p.perm[inst] = inst;
    }
    // ---- the working set: flagged rows in ascending order (rows 0..63 sit on wave 0, 64..127 on wave 1)
    if (wv == 0) ALMPC_STAMP(inst, 1);
    const bool flagged = tid < nz && y != 0.0;
    const unsigned long long fm = __ballot(flagged);
    if (lane == 0) s_cnt[wv] = __popcll(fm);
    __syncthreads();
    const int k0 = s_cnt[0] + s_cnt[1];
    {
        const int before = (wv == 0 ? 0 : s_cnt[0]) + __popcll(fm & ((1ull << lane) - 1ull));
        if (flagged && before < 64) s_rows[before] = tid;
    }
    __syncthreads();
    int32_t* rws = w.rows + (size_t)inst * 65;
    if (k0 <= 32 || k0 > 64) {   // (uniform) the finish's register mode takes such a set at once, or it is beyond the second tier
        if (tid == 0) rws[0] = 0;
        return;
    }
    // ---- K = G_i[W, W]: lane = position i, this wave's columns = positions c0 .. c0 + 15; (max, min) addressing: exactly symmetric
    if (wv == 0) ALMPC_STAMP(inst, 2);
    const double* Gi = w.G + (size_t)inst * w.G_stride;
    const int ri = lane < k0 ? s_rows[lane] : 0;
    gj16_row r;
    {   // (clamped addresses, no branch per element: all sixteen loads in flight before the first wait)
        int rj[16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) rj[jj] = s_rows[c0 + jj < k0 ? c0 + jj : 0];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int a = ri > rj[jj] ? ri : rj[jj], b = ri > rj[jj] ? rj[jj] : ri;
            r[jj] = Gi[(size_t)a * nzs + b];
        }
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) r[jj] = (c0 + jj < k0 && lane < k0) ? r[jj] : 0.0;
    }
    bool bad = false;
#ifdef ALMPC_STAMPS
    long long dbg_acc[2] = {0, 0};
#endif
    // two pivots per publication and barrier (gj16_pivot2); the last one of an odd set alone.  The loop over the pivots is unrolled by
    // 16 so that the register holding a pivot column is a compile-time index (see gj16_pivot)
    if (wv == 0) ALMPC_STAMP(inst, 3);
    int step = 0;
    for (int kb = 0; kb < k0 && !bad; kb += 16) {
        const bool own = c0 == kb;
        gj_static_for<0, 8>([&](auto jc) -> bool {
            constexpr int JJ = 2 * decltype(jc)::value;
            const int k = kb + JJ;
            if (k >= k0) return false;
            double* pw0 = s_prow[step & 1];
            double* pw1 = pw0 + 64;
            ++step;
            const bool two = k + 1 < k0;
            if (lane == k || (two && lane == k + 1)) {
                double* pw = lane == k ? pw0 : pw1;
#pragma unroll
                for (int jj = 0; jj < 16; jj += 2) *reinterpret_cast<double2*>(pw + c0 + jj) = make_double2(r[jj], r[jj + 1]);
            }
#ifdef ALMPC_STAMPS
            long long tq0 = __builtin_readcyclecounter();
#endif
            __syncthreads();
#ifdef ALMPC_STAMPS
            long long tq1 = __builtin_readcyclecounter();
            dbg_acc[0] += tq1 - tq0;
#endif
            double pj0[16], pj1[16];   // (every read of the pivot rows in one batch, in front of the pivot tests)
#pragma unroll
            for (int jj = 0; jj < 16; jj += 2) {
                const double2 t2 = *reinterpret_cast<const double2*>(pw0 + c0 + jj);
                pj0[jj] = t2.x; pj0[jj + 1] = t2.y;
            }
            const double col0 = pw0[lane], d11 = pw0[k];
#ifdef ALMPC_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            tq0 = __builtin_readcyclecounter();
            dbg_acc[1] += tq0 - tq1;
#endif
            if (two) {
#pragma unroll
                for (int jj = 0; jj < 16; jj += 2) {
                    const double2 t2 = *reinterpret_cast<const double2*>(pw1 + c0 + jj);
                    pj1[jj] = t2.x; pj1[jj + 1] = t2.y;
                }
                const double col1 = pw1[lane], d12 = pw0[k + 1], d22 = pw1[k + 1];
                const double i11 = fast_rcp_d(d11), t = d12 * i11, s22 = __builtin_fma(-t, d12, d22);
                if (!(d11 > 0.0) || !(s22 > 0.0)) { bad = true; return false; }   // (uniform: every wave reads the same pivots)
                if (c0 < k0) gj16_pivot2<JJ>(r, pj0, pj1, col0, col1, i11, t, s22, own, k, lane);   // (else: columns beyond the set; the wave still meets the barriers)
            } else {
                if (!(d11 > 0.0)) { bad = true; return false; }
                if (c0 < k0) gj16_pivot<JJ>(r, pj0, col0, d11, own, k, lane);
            }
            return true;
        });
    }
    if (bad) {
        if (tid == 0) rws[0] = 0;
        return;
    }
    // ---- out: column j of the 64 x 64 second-tier matrix, identity beyond the set
    if (wv == 0) ALMPC_STAMP(inst, 4);
#ifdef ALMPC_STAMPS
    if (g_stamps && tid == 0) { g_stamps[(size_t)inst * 16 + 13] = dbg_acc[0]; g_stamps[(size_t)inst * 16 + 14] = dbg_acc[1]; }
#endif
    double* out = w.sinv + (size_t)inst * 64 * 64;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int j = c0 + jj;
        out[(size_t)j * 64 + lane] = (j < k0 && lane < k0) ? -r[jj] : (lane == j ? 1.0 : 0.0);
    }
    if (wv == 0) {
        if (lane < k0) rws[1 + lane] = ri;
        if (lane == 0) rws[0] = k0;
    }
    if (wv == 0) ALMPC_STAMP(inst, 5);
#ifdef ALMPC_STAMPS
    if (g_stamps && tid == 0) g_stamps[(size_t)inst * 16 + 7] = __builtin_amdgcn_s_memrealtime();
#endif
}

// k_guess_shift: the same hand-off for a receding-horizon step with per-instance models (almpc_relin_fnn_step, warm_start = 1): the
// guess is the previous step's input trajectory shifted by one stage (stage k <- stage k+1, the last stage repeated) -- the classic
// MPC warm start.  z = that point in the new scaled coordinates, clipped to the box; working set = its rows on a bound.  The finish
// is exact whatever its start; what goes away is the ADMM phase and the KKT inverse of the design.  uprev: [batch][N][m].
inline __global__ __launch_bounds__(256) void k_guess_shift(AdmmInstParams p, const double* uprev, int N) {
    const int nz = p.nz, nzs = p.nzs, n = p.n, m = p.m, lane = threadIdx.x & 63;
    const int inst = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (inst >= p.batch) return;
    const double* Vi = p.Vs + (size_t)inst * n * nzs;
    const double* dv = p.dvec + (size_t)inst * nzs;
    const size_t o = (size_t)inst * nzs;
    for (int r = lane; r < nzs; r += 64) {
        double v = 0.0, z = 0.0, y = 0.0;
        if (r < nz) {
            v = p.v0S[(size_t)inst * nz + r];
            for (int c = 0; c < n; ++c)
                v += Vi[(size_t)c * nzs + r] * (p.x0[(size_t)inst * n + c] - p.xref[(size_t)inst * p.xref_stride + c]);
            const int k = r / m, a = r % m;
            const double up = uprev[(size_t)inst * nz + (size_t)(k + 1 < N ? k + 1 : N - 1) * m + a];
            const double ur = p.uref[(size_t)inst * p.uref_stride + r], di = 1.0 / dv[r];
            const double lo = (p.umin[a] - ur) * di, hi = (p.umax[a] - ur) * di;  // as k_admm_inst forms them
            z = fmin(fmax((up - ur) * di, lo), hi);
            y = (z >= hi) ? 1.0 : ((z <= lo) ? -1.0 : 0.0);
        }
        p.xs[o + r] = z; p.zs[o + r] = z; p.ys[o + r] = y; p.v0[o + r] = v;
    }
    if (lane == 0) {
        p.iters[inst] = 0;
        p.status[inst] = 1;   // "not converged": the finish sets 0 when it certifies the optimum
        p.piters[inst] = 0;
        p.perm[inst] = inst;
    }
}

// ------------------------------------------------------------------------------------------------
// k_design_instance: condensed Hessian and gradient matrix of ONE instance per workgroup, entirely in LDS.
// The dense route (k_design_gamma + k_design_hessian) materialises Gamma_i and Qbar Gamma_i in HBM (2 x 393 KB per quadrotor-size
// instance) and contracts them with MFMA; for a batch of instances that traffic, not the flops, is the cost.  Gamma is block
// lower-triangular Toeplitz with blocks G_k = A^k B, so with Q G_k, P G_k, Phi_k = A^k in LDS
//     H_ij = 2 [ sum_{a=0}^{N-2-i} G_a' (Q G_{a+d}) + G_{N-1-i}' (P G_{N-1-i+d}) ],  d = i - j >= 0      (m x m blocks)
//     F_i  = 2 [ sum_{a=0}^{N-2-i} E_a + E^P_{N-1-i} ] Phi_{i+1},   E_a = G_a' Q A^a,  E^P_a = G_a' P A^a   (m x n blocks)
// i.e. prefix sums along the block diagonals: O(N^2 m^2 n) flops instead of O(N^3 m^2 n), nothing but A_i, B_i, P_i read and
// H_i, F_i written.  Same H and F as k_design_hessian up to summation order (R, S terms as there: src/sub/design_mpc.jl:405-468).
// Q and P symmetric (the host symmetrises them).  LDS: (N+1) n^2 + 5 N n m + 4 n^2 + ... doubles; the host falls back to the
// dense route when that does not fit.
// ------------------------------------------------------------------------------------------------
struct DesignInstParams {
    int n, m, N, nz, useR, useS;
    const double* A; const double* B; const double* P; long sA, sB, sP;   // per instance
    const double* Q; const double* R; const double* S;                     // shared
    double* H; double* F; long sH, sF;                                     // column-major nz x nz, nz x n
    int* flag = nullptr; long sFlag = 0;                                   // design flag of the instance: cleared here when given (instead of
                                                                           // by a memset launch in front of every re-design)
    // the Jacobi scaling as the tail of this kernel (design_scale_body on the H_i, F_i just written) when Hs is given: d_i, H'_i, F'_i
    double* d = nullptr; double* Hs = nullptr; double* Fs = nullptr; long sd = 0, sHs = 0, sFs = 0; int nzs = 0;
    // the re-linearisation pipeline: the instance's own workgroup linearises its model first (fnn_jacobian_point by wave 0 into A, B,
    // which are then read back as before) when fnn_on; network weights + one wave's scratch at smem + fnn_off
    int fnn_on = 0; size_t fnn_off = 0; FnnParams fnn;
};

__host__ __device__ inline size_t design_instance_lds_doubles(int n, int m, int N) {
    return (size_t)4 * n * n + (size_t)n * m + (size_t)(N + 1) * n * n + (size_t)5 * N * n * m;
}

// <NC, MC>: n and m at compile time for the common shapes (0, 0: run time): the products are short loops over n / m with two LDS
// reads per term, which only unroll -- and only then have their reads in flight together -- when the bounds are constants
// (same finding as k_riccati_t).
template <int NC, int MC>
__global__ __launch_bounds__(256) void k_design_instance_t(DesignInstParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n = NC ? NC : p.n, m = MC ? MC : p.m, N = p.N, nz = p.nz;
    const int nn = n * n, nm = n * m;
    const double* A = p.A + blockIdx.x * p.sA;
    const double* B = p.B + blockIdx.x * p.sB;
    const double* P = p.P + blockIdx.x * p.sP;
    double* H = p.H + blockIdx.x * p.sH;
    double* F = p.F + blockIdx.x * p.sF;
    double* As = smem;             // [n][n] column-major: As[j*n+i] = A_ij
    double* Qs = As + nn;
    double* Ps = Qs + nn;
    double* tmp = Ps + nn;         // [n][n] scratch
    double* Bs = tmp + nn;         // [n][m] column-major
    double* Phi = Bs + nm;         // [N+1][n*n]: Phi[k] = A^k
    double* Gk = Phi + (size_t)(N + 1) * nn;   // [N][n*m]: G_k = A^k B
    double* QG = Gk + (size_t)N * nm;          // Q G_k
    double* PG = QG + (size_t)N * nm;          // P G_k
    double* E = PG + (size_t)N * nm;           // [N][m*n]: E_a (then prefix sums), element (p, j) at [j*m + p]
    double* EP = E + (size_t)N * nm;           // E^P_a
    const int T = blockDim.x;
    if (p.fnn_on) {
        double* wsm = smem + p.fnn_off;
        fnn_stage_weights(p.fnn, wsm);
        __syncthreads();
        if (threadIdx.x < 64)
            fnn_jacobian_point(p.fnn, blockIdx.x, threadIdx.x, wsm, wsm + fnn_weights_doubles(p.fnn.n, p.fnn.m, p.fnn.H, p.fnn.L));
        __threadfence_block();
        __syncthreads();   // (A_i, B_i are in the handle's model slots now: read back below, by the step's rollout and by the redo)
    }
    if (p.flag && threadIdx.x == 0) p.flag[blockIdx.x * p.sFlag] = 0;
    for (int t = threadIdx.x; t < nn; t += T) { As[t] = A[t]; Qs[t] = p.Q[t]; Ps[t] = P[t]; Phi[t] = (t % n == t / n) ? 1.0 : 0.0; }
    for (int t = threadIdx.x; t < nm; t += T) Bs[t] = B[t];
    __syncthreads();
    for (int k = 1; k <= N; ++k) {  // Phi[k] = A Phi[k-1]
        const double* prev = Phi + (size_t)(k - 1) * nn;
        double* cur = Phi + (size_t)k * nn;
        for (int t = threadIdx.x; t < nn; t += T) {
            const int i = t % n, j = t / n;
            double s = 0.0;
            for (int l = 0; l < n; ++l) s += As[l * n + i] * prev[j * n + l];
            cur[t] = s;
        }
        __syncthreads();
    }
    for (int t = threadIdx.x; t < N * nm; t += T) {  // G_k = Phi[k] B
        const int k = t / nm, e = t % nm, i = e % n, j = e / n;
        const double* ph = Phi + (size_t)k * nn;
        double s = 0.0;
        for (int l = 0; l < n; ++l) s += ph[l * n + i] * Bs[j * n + l];
        Gk[t] = s;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < N * nm; t += T) {  // Q G_k, P G_k
        const int k = t / nm, e = t % nm, i = e % n, j = e / n;
        const double* g = Gk + (size_t)k * nm + (size_t)j * n;
        double sq = 0.0, sp = 0.0;
        for (int l = 0; l < n; ++l) { sq += Qs[l * n + i] * g[l]; sp += Ps[l * n + i] * g[l]; }
        QG[t] = sq; PG[t] = sp;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < N * nm; t += T) {  // E_a[p][j] = sum_s (Q G_a)[s][p] Phi[a][s][j]  (Q symmetric)
        const int a = t / nm, e = t % nm, pp = e % m, j = e / m;
        const double* ph = Phi + (size_t)a * nn + (size_t)j * n;
        const double* qg = QG + (size_t)a * nm + (size_t)pp * n;
        const double* pg = PG + (size_t)a * nm + (size_t)pp * n;
        double sq = 0.0, sp = 0.0;
        for (int l = 0; l < n; ++l) { sq += qg[l] * ph[l]; sp += pg[l] * ph[l]; }
        E[t] = sq; EP[t] = sp;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nm; e += T)  // prefix sums over a
        for (int a = 1; a < N; ++a) E[(size_t)a * nm + e] += E[(size_t)(a - 1) * nm + e];
    __syncthreads();
    // F_i = 2 (SE[N-2-i] + EP[N-1-i]) Phi[i+1]
    for (int t = threadIdx.x; t < N * nm; t += T) {
        const int i = t / nm, e = t % nm, pp = e % m, j = e / m;
        const double* ph = Phi + (size_t)(i + 1) * nn + (size_t)j * n;
        const double* se = (i <= N - 2) ? E + (size_t)(N - 2 - i) * nm : nullptr;
        const double* ep = EP + (size_t)(N - 1 - i) * nm;
        double s = 0.0;
        for (int l = 0; l < n; ++l) s += ((se ? se[l * m + pp] : 0.0) + ep[l * m + pp]) * ph[l];
        F[(size_t)j * nz + i * m + pp] = 2.0 * s;
    }
    // H: thread (d, p, q) walks block diagonal d from the last block row up, carrying the prefix sum of the Q terms
    const int mm = m * m;
    for (int t = threadIdx.x; t < N * mm; t += T) {
        const int d = t / mm, e = t % mm, pp = e % m, qq = e / m;
        double C = 0.0;
        // Block diagonal d is walked from the last block row up (i = N-1 .. d), skewed by d steps: at step tau every thread is in
        // block COLUMN j = N-1-tau, so the lanes (d, pp) of one qq store a contiguous piece of one column of H (the plain walk had
        // every lane in a different column: 8-byte scattered stores, 0.36 TB/s)
        for (int tau = 0; tau < N; ++tau) {   // (uniform trip count: the skew is in the predicate)
            if (tau < d) continue;
            const int i = N - 1 - tau + d;
            if (i < N - 1) {
                const int a = N - 2 - i;
                const double* g = Gk + (size_t)a * nm + (size_t)pp * n;
                const double* qg = QG + (size_t)(a + d) * nm + (size_t)qq * n;
                double s = 0.0;
                for (int l = 0; l < n; ++l) s += g[l] * qg[l];
                C += s;
            }
            const double* g = Gk + (size_t)(N - 1 - i) * nm + (size_t)pp * n;
            const double* pg = PG + (size_t)(N - 1 - i + d) * nm + (size_t)qq * n;
            double sp = 0.0;
            for (int l = 0; l < n; ++l) sp += g[l] * pg[l];
            double v = 2.0 * (C + sp);
            const int j = i - d;
            if (p.useR && d == 0) v += 2.0 * p.R[(size_t)qq * m + pp];
            if (p.useS) {  // delta_u[:,i] = u[:,i] - u[:,i+1], i = 1..N-1 (src/sub/design_mpc.jl:429-431)
                if (d == 0) v += 2.0 * ((i <= N - 2 ? 1 : 0) + (i >= 1 ? 1 : 0)) * p.S[(size_t)qq * m + pp];
                else if (d == 1) v -= 2.0 * p.S[(size_t)qq * m + pp];
            }
            const int row = i * m + pp, col = j * m + qq;
            H[(size_t)col * nz + row] = v;
            if (d > 0) H[(size_t)row * nz + col] = v;
        }
    }
    if (p.Hs) {   // scaling of what this workgroup has just written (L2 / L1 resident), flag needed: p.flag
        __threadfence_block();
        __syncthreads();   // (every LDS buffer above is dead now: the scaling's 128 doubles sit at the start; the host checks the size)
        design_scale_body(nz, p.nzs, n, H, F, p.d + blockIdx.x * p.sd, p.Hs + blockIdx.x * p.sHs, p.Fs + blockIdx.x * p.sFs,
                          p.flag + blockIdx.x * p.sFlag, 1, smem);
    }
}

// ------------------------------------------------------------------------------------------------
// k_design_ltv: time-varying models (SQP / multiple shooting around the QP engine, BASELINE.json configs[4]): stage k of instance
// i has its own (A_ik, B_ik) and a defect c_ik, so with dx_0 = 0
//     dx_{k+1} = A_k dx_k + B_k v_k + c_k   =>   dX = Gamma~ v + g~,   Gamma~ block (k, j) = A_{k-1} ... A_{j+1} B_j,  j < k.
// The QP in v has H = 2 (Gamma~' Qbar Gamma~ + Rbar + D'Sbar D) and state part of the gradient q = 2 Gamma~' Qbar (g~ + ebar),
// ebar_k = xbar_k - x_ref_k.  Gamma~ is not Toeplitz, but only its CURRENT row block (n x nz) is ever needed: one workgroup
// per instance walks the stages, propagates the row block and g~ (one n x n product each) and accumulates H and q in LDS:
// O(nz^2 n N / 3) flops, H_i and q_i are the only things written.  The R, S part of the gradient is added from qadd.
// ------------------------------------------------------------------------------------------------
// ---- SQP outer loop: parameters of its glue kernels (csrc/almpc_sqp.hip.h) and the part that rides along in k_design_ltv_reg ----
struct SqpParams {
    int n, m, N, nz, batch, useR, useS;
    const double* xref;   // [(N+1)][n] shared state reference
    const double* uref;   // [N][m]     shared input reference
    const double* R; const double* S;   // m x m, symmetrised
    const double* umin; const double* umax;
    double* xbar;         // [batch][(N+1)][n]
    double* ubar;         // [batch][N][m]
    const double* fval;   // [batch][N][n]  network outputs at (xbar_k, ubar_k)
    const double* A; const double* B;   // [batch][N][n*n], [batch][N][n*m]
    double* c;            // [batch][N][n]  defects f(xbar_k, ubar_k) - xbar_{k+1}
    double* ebar;         // [batch][N][n]  xbar_{k+1} - xref_{k+1}
    double* qadd;         // [batch][nz]    2 Rbar (ubar - uref) + 2 D'Sbar D ubar
    const double* v;      // [batch][N][m]  QP solution (the step kernels' e_u output); overwritten with ubar - uref
    const int* flag;      // [batch] design failure of this iteration (non-zero: skip the update)
    const int* status;    // [batch] status of the QP solve (2 = non-finite: skip the update)
    int* bad;             // [batch] sticky: some iteration of this instance was skipped
    unsigned long long* stats;  // [2]: bit patterns of max |v| and max |c| over the batch (non-negative doubles order like integers)
    double step_scale;
    // step rule 1 (merit-function safeguard): see k_sqp_prepare
    int adaptive;
    double mu;            // weight of the defects in the merit function J + mu |defects|_1
    const double* Q; const double* P; long sP;   // cost weights for J (P shared or per instance)
    double* mer;          // [batch][4]: step factor a, merit of the last accepted point, redo flag (1: this iteration is void), spare
    double* xback; double* uback;   // [batch][(N+1) n], [batch][nz]: last accepted point
    double* dxback; double* vback;  // its step (dx of every stage, v), so that a rejected trial can be re-taken shorter
    double *x, *ex, *u, *eu;    // result buffers: the iterate after the update
};

// Before the QP: defects, state errors and the input part of the gradient, one workgroup per instance.
// Step rule 1.  Full Gauss-Newton steps are not globally convergent (3 of the 256 benchmark instances end in a cycle), and
// heuristics on |v| alone misfire in the first iterations, where growing steps are normal.  The safeguard is the classical l1
// merit function phi = J(x, u) + mu |f(x, u) - x+|_1, evaluated a posteriori: the network outputs at the point reached by the last
// step are computed by THIS iteration's linearisation anyway, so the test costs one reduction.  If phi did not decrease, the
// point is rejected: the iterate goes back to the last accepted point plus HALF the step (both kept), this iteration's QP -- built
// at the rejected point -- is void for the instance (redo flag: k_sqp_step leaves it alone), and the next iteration tests the
// shorter step.  Accepted steps double the factor back up to 1; at 1/64 a step is accepted regardless.
// A device function: k_sqp_prepare is one caller, the head of k_design_ltv_reg the other (DesignLtvParams::prep_on: one launch less per
// iteration; the first 256 threads of that kernel's workgroup do exactly what the kernel's 256 did).
__device__ inline void sqp_prepare_body(const SqpParams& p, const size_t i) {
    const int n = p.n, m = p.m, N = p.N, nz = p.nz;
    const int tid = threadIdx.x < 256 ? (int)threadIdx.x : 0x3fffffff, nthr = 256;   // (threads beyond 256 only meet the barriers)
    double* xbw = p.xbar + i * (size_t)(N + 1) * n;
    double* ubw = p.ubar + i * (size_t)nz;
    if (p.adaptive) {
        __shared__ double red[8];
        __shared__ int reject;
        double part = 0.0;
        const double* Pm = p.P + i * p.sP;
        for (int t = tid; t < (N + 1) * n; t += nthr) {   // e_x' W e_x, W = Q for stages 1..N, P for N+1
            const int k = t / n, r = t % n;
            const double* W = (k == N) ? Pm : p.Q;
            double sdot = 0.0;
            for (int j = 0; j < n; ++j) sdot += W[(size_t)j * n + r] * (xbw[k * n + j] - p.xref[k * n + j]);
            part += (xbw[t] - p.xref[t]) * sdot;
        }
        for (int t = tid; t < nz; t += nthr) {
            const int k = t / m, a = t % m;
            if (p.useR) {
                double sdot = 0.0;
                for (int c2 = 0; c2 < m; ++c2) sdot += p.R[(size_t)c2 * m + a] * (ubw[k * m + c2] - p.uref[k * m + c2]);
                part += (ubw[t] - p.uref[t]) * sdot;
            }
            if (p.useS && k + 1 < N) {
                double sdot = 0.0;
                for (int c2 = 0; c2 < m; ++c2) sdot += p.S[(size_t)c2 * m + a] * (ubw[k * m + c2] - ubw[(k + 1) * m + c2]);
                part += (ubw[t] - ubw[t + m]) * sdot;
            }
        }
        for (int t = tid; t < N * n; t += nthr) part += p.mu * fabs(p.fval[i * (size_t)N * n + t] - xbw[n + t]);
        part = wave_sum(part);
        if ((threadIdx.x & 63) == 0 && threadIdx.x < 256) red[threadIdx.x >> 6] = part;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double phi = (red[0] + red[1]) + (red[2] + red[3]);
            double a = p.mer[4 * i], ref = p.mer[4 * i + 1];
            // (a non-finite phi is rejected like an increase; the first iteration after `start` has ref = +inf)
            const bool ok = (phi <= ref + 1e-12 * fabs(ref) + 1e-300) || a <= 1.0 / 64.0;
            if (ok) { ref = phi; a = fmin(1.0, 2.0 * a); }
            else a *= 0.5;
            p.mer[4 * i] = a; p.mer[4 * i + 1] = ref; p.mer[4 * i + 2] = ok ? 0.0 : 1.0;
            red[4] = a;
            reject = ok ? 0 : 1;
        }
        __syncthreads();
        if (reject) {  // back to the last accepted point plus the shorter step
            const double a = red[4];
            for (int t = tid; t < (N + 1) * n; t += nthr)
                xbw[t] = p.xback[i * (size_t)(N + 1) * n + t] + a * p.dxback[i * (size_t)(N + 1) * n + t];
            for (int t = tid; t < nz; t += nthr) {
                const int am = t % m;
                ubw[t] = fmin(fmax(p.uback[i * (size_t)nz + t] + a * p.vback[i * (size_t)nz + t], p.umin[am]), p.umax[am]);
            }
            __syncthreads();
        }
    }
    const double* xb = xbw;
    const double* ub = ubw;
    for (int t = tid; t < N * n; t += nthr) {
        const double xn = xb[n + t];  // stage k+1, component j  (t = k*n + j)
        p.c[i * (size_t)N * n + t] = p.fval[i * (size_t)N * n + t] - xn;
        p.ebar[i * (size_t)N * n + t] = xn - p.xref[n + t];
    }
    for (int t = tid; t < nz; t += nthr) {
        const int k = t / m, a = t % m;
        double g = 0.0;
        if (p.useR) {
            double s = 0.0;
            for (int c2 = 0; c2 < m; ++c2) s += p.R[(size_t)c2 * m + a] * (ub[k * m + c2] - p.uref[k * m + c2]);
            g += 2.0 * s;
        }
        if (p.useS) {  // the input-rate cost is on u itself (src/sub/design_mpc.jl:423-446): row t of 2 D'Sbar D ubar
            double s = 0.0;
            for (int c2 = 0; c2 < m; ++c2) {
                const double sac = p.S[(size_t)c2 * m + a];
                if (k + 1 < N) s += sac * (ub[k * m + c2] - ub[(k + 1) * m + c2]);
                if (k > 0) s -= sac * (ub[(k - 1) * m + c2] - ub[k * m + c2]);
            }
            g += 2.0 * s;
        }
        p.qadd[i * (size_t)nz + t] = g;
    }
}


struct DesignLtvParams {
    int n, m, N, nz, useR, useS;
    const double* A; const double* B;        // [batch][N][n*n], [batch][N][n*m]  (column-major blocks)
    const double* c; const double* ebar;     // [batch][N][n] defects / state errors at the linearisation (stages 1..N); nullable
    const double* P; long sP;                // terminal weight, shared (sP = 0) or per instance
    const double* Q; const double* R; const double* S;
    const double* qadd;                      // [batch][nz] input part of the gradient (host), nullable
    double* H; double* q;                    // [batch][nz*nz] column-major, [batch][nz]
    // k_design_ltv_reg only, sc_Hs given: the Jacobi scaling as the kernel's tail (design_scale_body on the H_i just written: d_i, the
    // symmetrised H'_i; the design flag cleared at the start and set there) and the scaled gradient fS_i = d_i .* q_i -- the SQP
    // iteration's k_design_scale, k_fs_scale and flag memset (three launches) ride along
    double* sc_d = nullptr; double* sc_Hs = nullptr; double* sc_fS = nullptr; int* sc_flag = nullptr; int nzs = 0;
    // k_design_ltv_reg only, prep_on: the SQP iteration's k_sqp_prepare (defects c, state errors ebar, input gradient qadd, the merit
    // test of the last step) as the kernel's HEAD -- its outputs are this kernel's inputs
    int prep_on = 0;
    SqpParams prep;
};

__host__ __device__ inline size_t design_ltv_lds_doubles(int n, int m, int N) {
    const size_t nz = (size_t)m * N;
    return nz * nz + 3 * (size_t)n * nz + 4 * (size_t)n * n + (size_t)n * m + 3 * (size_t)n + nz;
}

inline __global__ __launch_bounds__(256) void k_design_ltv(DesignLtvParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n = p.n, m = p.m, N = p.N, nz = p.nz, nn = n * n, nm = n * m;
    const size_t inst = blockIdx.x;
    double* Hs = smem;                 // [nz][nz] column-major accumulator
    double* Gc = Hs + (size_t)nz * nz; // [nz][n]: current row block of Gamma~, element (p, c) at c*n + p
    double* Gn = Gc + (size_t)n * nz;
    double* T = Gn + (size_t)n * nz;   // Q_k times the row block
    double* Ak = T + (size_t)n * nz;
    double* Qs = Ak + nn;
    double* Ps = Qs + nn;
    double* Bk = Ps + nn;
    double* gk = Bk + nm;              // g~_k
    double* gn = gk + n;
    double* rv = gn + n;               // g~ + ebar of the current stage
    double* qs = rv + n;               // [nz]
    const int T_ = blockDim.x;
    for (int t = threadIdx.x; t < nz * nz; t += T_) Hs[t] = 0.0;
    for (int t = threadIdx.x; t < n * nz; t += T_) { Gc[t] = 0.0; Gn[t] = 0.0; }
    for (int t = threadIdx.x; t < nz; t += T_) qs[t] = 0.0;
    for (int t = threadIdx.x; t < nn; t += T_) { Qs[t] = p.Q[t]; Ps[t] = p.P[inst * p.sP + t]; }
    for (int t = threadIdx.x; t < n; t += T_) gk[t] = 0.0;
    __syncthreads();
    for (int k = 0; k < N; ++k) {
        const double* Ag = p.A + (inst * N + k) * nn;
        const double* Bg = p.B + (inst * N + k) * nm;
        for (int t = threadIdx.x; t < nn; t += T_) Ak[t] = Ag[t];
        for (int t = threadIdx.x; t < nm; t += T_) Bk[t] = Bg[t];
        __syncthreads();
        const int wcols = (k + 1) * m;  // columns of the row block that are non-zero after this stage
        for (int t = threadIdx.x; t < n * wcols; t += T_) {
            const int pr = t % n, c = t / n;
            double v;
            if (c >= k * m) v = Bk[(c - k * m) * n + pr];
            else {
                v = 0.0;
                for (int l = 0; l < n; ++l) v += Ak[l * n + pr] * Gc[c * n + l];
            }
            Gn[t] = v;
        }
        for (int pr = threadIdx.x; pr < n; pr += T_) {
            double v = p.c ? p.c[(inst * N + k) * n + pr] : 0.0;
            for (int l = 0; l < n; ++l) v += Ak[l * n + pr] * gk[l];
            gn[pr] = v;
        }
        __syncthreads();
        { double* t = Gc; Gc = Gn; Gn = t; }
        { double* t = gk; gk = gn; gn = t; }
        const double* Qk = (k == N - 1) ? Ps : Qs;  // stage N+1 carries only P (src/sub/design_mpc.jl:448-456)
        for (int t = threadIdx.x; t < n * wcols; t += T_) {
            const int pr = t % n, c = t / n;
            double v = 0.0;
            for (int l = 0; l < n; ++l) v += Qk[l * n + pr] * Gc[c * n + l];
            T[t] = v;
        }
        for (int pr = threadIdx.x; pr < n; pr += T_) rv[pr] = gk[pr] + (p.ebar ? p.ebar[(inst * N + k) * n + pr] : 0.0);
        __syncthreads();
        for (int t = threadIdx.x; t < wcols * wcols; t += T_) {
            const int c1 = t % wcols, c2 = t / wcols;
            double v = 0.0;
            for (int l = 0; l < n; ++l) v += Gc[c1 * n + l] * T[c2 * n + l];
            Hs[(size_t)c2 * nz + c1] += v;
        }
        for (int c = threadIdx.x; c < wcols; c += T_) {
            double v = 0.0;
            for (int l = 0; l < n; ++l) v += T[c * n + l] * rv[l];
            qs[c] += v;
        }
        __syncthreads();
    }
    double* H = p.H + inst * (size_t)nz * nz;
    for (int t = threadIdx.x; t < nz * nz; t += T_) {
        const int r = t % nz, c = t / nz;
        double v = 2.0 * Hs[t];
        const int ir = r / m, ar = r % m, ic = c / m, ac = c % m;
        if (p.useR && ir == ic) v += 2.0 * p.R[(size_t)ac * m + ar];
        if (p.useS) {  // delta_u[:,i] = u[:,i] - u[:,i+1], i = 1..N-1 (src/sub/design_mpc.jl:429-431)
            if (ir == ic) v += 2.0 * ((ir <= N - 2 ? 1 : 0) + (ir >= 1 ? 1 : 0)) * p.S[(size_t)ac * m + ar];
            else if (ir - ic == 1 || ic - ir == 1) v -= 2.0 * p.S[(size_t)ac * m + ar];
        }
        H[t] = v;
    }
    for (int t = threadIdx.x; t < nz; t += T_) p.q[inst * nz + t] = 2.0 * qs[t] + (p.qadd ? p.qadd[inst * nz + t] : 0.0);
}

// Register-accumulator build of the same design for nz <= 128: 1024 threads, thread (tx, ty) keeps the 4 x 4 strided tile
// H[tx + 32 i][ty + 32 j] in registers for the whole walk, so a stage costs 8 n LDS reads per thread instead of a read-modify-write
// of LDS per element (the LDS build above spends 85 % of its time there), the row blocks are stored stage-row-major ([l][c]:
// lanes read consecutive addresses), q sits in the registers of threads c < nz, the next stage's (A, B) are fetched one stage
// two stages ahead (a stage is shorter than an HBM round trip), defects and state errors are staged in LDS once, and a stage
// needs two barriers -- ONE when n is a template parameter (round 3): thread c then builds column c of the new row block AND of Q_k
// times it in registers (2 n^2 FMAs) and writes both, double buffered, before the stage's only barrier; 16 waves meet at a barrier
// in ~450 cycles, the stage's arithmetic is ~400.  R and S sit in LDS (the epilogue's loads of them, one per element behind a
// branch, were waited for one by one).  LDS: 4 n 128 + 4 n^2 + 2 n m + 4 n + 2 N n + 2 m^2 doubles (20 KB for n = 4, N = 50).
constexpr int LTV_REG_NZ = 128;  // (the kernel uses t >> 7 and t & 127)

__host__ __device__ inline size_t design_ltv_reg_lds_doubles(int n, int m, int N) {
    return 4 * (size_t)n * LTV_REG_NZ + 4 * (size_t)n * n + 2 * (size_t)n * m + 4 * (size_t)n + 2 * (size_t)N * n + 2 * (size_t)m * m;
}

// TS: side of a thread's register tile (4: 1024 threads, the default; 8: 256 threads, half the LDS bytes per FMA -- measured slower, 124
// against 81 us for N = 50: a stage (1.6 us) is a chain of dependent LDS steps that 16 waves hide better than four).
template <int NC, int TS = 4>  // NC > 0: the state dimension at compile time (the stage loops unroll); 0: any n
__global__ __launch_bounds__((128 / TS) * (128 / TS)) void k_design_ltv_reg(DesignLtvParams p) {
    constexpr int DX = 128 / TS;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NZP = LTV_REG_NZ;
    const int n = NC > 0 ? NC : p.n, m = p.m, N = p.N, nz = p.nz, nn = n * n, nm = n * m, nab = nn + nm;
    const size_t inst = blockIdx.x;
    if (p.prep_on) {   // the SQP iteration's defects, state errors and input gradient (this kernel's inputs) and its merit test
        sqp_prepare_body(p.prep, inst);
        __syncthreads();
    }
    double* Gc = smem;                    // [n][NZP] current row block of Gamma~ (row l of the block, column c)
    double* Gn = Gc + (size_t)n * NZP;
    double* T = Gn + (size_t)n * NZP;     // Q_k times the row block
    double* T2 = T + (size_t)n * NZP;     // (second buffer of the one-barrier path)
    double* AB = T2 + (size_t)n * NZP;    // [2][nn + nm]: (A_k, B_k) of the current and of the next stage
    double* Qs = AB + 2 * nab;
    double* Ps = Qs + nn;
    double* gk = Ps + nn;
    double* gn = gk + n;
    double* rv2 = gn + n;                 // [2][n]: g~ + ebar of the current stage, double buffered over the stages
    double* cs = rv2 + 2 * n;             // [N][n] defects
    double* es = cs + (size_t)N * n;      // [N][n] state errors
    double* Rs = es + (size_t)N * n;      // [m][m]
    double* Ss = Rs + (size_t)m * m;      // [m][m]
    const int tid = threadIdx.x, tx = tid % DX, ty = tid / DX, T_ = DX * DX;
    double acc[TS][TS];
#pragma unroll
    for (int i = 0; i < TS; ++i)
#pragma unroll
        for (int j = 0; j < TS; ++j) acc[i][j] = 0.0;
    double qacc = 0.0;
    // (A, B) of the stages are contiguous per instance: [N][nn] and [N][nm]; element e < nab of stage k
    const double* Ag = p.A + inst * N * (size_t)nn;
    const double* Bg = p.B + inst * N * (size_t)nm;
    auto ab_load = [&](int k) -> double { return tid < nn ? Ag[(size_t)k * nn + tid] : Bg[(size_t)k * nm + tid - nn]; };
    double pre = 0.0;  // stage k + 2's element, in flight for a whole stage before it is written to LDS
    if (tid < nab) {
        AB[tid] = ab_load(0);
        if (N > 1) AB[nab + tid] = ab_load(1);
        if (N > 2) pre = ab_load(2);
    }
    for (int t = tid; t < 4 * n * NZP; t += T_) Gc[t] = 0.0;  // Gc, Gn, T, T2
    for (int t = tid; t < nn; t += T_) { Qs[t] = p.Q[t]; Ps[t] = p.P[inst * p.sP + t]; }
    for (int t = tid; t < m * m; t += T_) { Rs[t] = p.useR ? p.R[t] : 0.0; Ss[t] = p.useS ? p.S[t] : 0.0; }
    for (int t = tid; t < n; t += T_) gk[t] = 0.0;
    for (int t = tid; t < N * n; t += T_) {
        cs[t] = p.c ? p.c[inst * N * n + t] : 0.0;
        es[t] = p.ebar ? p.ebar[inst * N * n + t] : 0.0;
    }
    __syncthreads();
    for (int k = 0; k < N; ++k) {
        const double* Ak = AB + (k & 1) * nab;
        const double* Bk = Ak + nn;
        const int wcols = (k + 1) * m;  // columns of the row block that are non-zero after this stage
        const double* Qk = (k == N - 1) ? Ps : Qs;  // stage N+1 carries only P (src/sub/design_mpc.jl:448-456)
        if constexpr (NC > 0) {
            // one barrier per stage: thread c owns column c of the new row block (n values in registers) and of Q_k times it
            double* Tn = (k & 1) ? T2 : T;
            if (tid < wcols) {
                const int c = tid;
                double g[NC];
                if (c >= k * m) {
#pragma unroll
                    for (int pr = 0; pr < NC; ++pr) g[pr] = Bk[(c - k * m) * NC + pr];
                } else {
                    double gc[NC];
#pragma unroll
                    for (int l = 0; l < NC; ++l) gc[l] = Gc[l * NZP + c];
#pragma unroll
                    for (int pr = 0; pr < NC; ++pr) {
                        double v = 0.0;
#pragma unroll
                        for (int l = 0; l < NC; ++l) v += Ak[l * NC + pr] * gc[l];
                        g[pr] = v;
                    }
                }
#pragma unroll
                for (int pr = 0; pr < NC; ++pr) {
                    Gn[pr * NZP + c] = g[pr];
                    double v = 0.0;
#pragma unroll
                    for (int l = 0; l < NC; ++l) v += Qk[l * NC + pr] * g[l];
                    Tn[pr * NZP + c] = v;
                }
            }
            if (tid >= T_ - n) {  // the last n threads (for n <= 7 they have no share of the row block)
                const int pr = tid - (T_ - n);
                double v = cs[k * n + pr];
                for (int l = 0; l < n; ++l) v += Ak[l * n + pr] * gk[l];
                gn[pr] = v;
                rv2[(k & 1) * n + pr] = v + es[k * n + pr];
            }
            __syncthreads();
            { double* t = Gc; Gc = Gn; Gn = t; }
            { double* t = gk; gk = gn; gn = t; }
            const double* rv = rv2 + (k & 1) * n;
            if (tid < nab) {  // this stage's (A, B) slot is free (last read before the barrier above): it takes stage k + 2
                if (k + 2 < N) AB[(k & 1) * nab + tid] = pre;
                if (k + 3 < N) pre = ab_load(k + 3);
            }
            const int ni = (wcols + DX - 1) / DX;  // DX-column groups that hold non-zeros
#pragma unroll
            for (int l = 0; l < NC; ++l) {
                double a[TS], b[TS];
#pragma unroll
                for (int i = 0; i < TS; ++i) {
                    a[i] = i < ni ? Gc[l * NZP + tx + DX * i] : 0.0;
                    b[i] = i < ni ? Tn[l * NZP + ty + DX * i] : 0.0;
                }
#pragma unroll
                for (int i = 0; i < TS; ++i)
#pragma unroll
                    for (int j = 0; j < TS; ++j) acc[i][j] += a[i] * b[j];
            }
            if (tid < wcols) {
                double v = 0.0;
#pragma unroll
                for (int l = 0; l < NC; ++l) v += Tn[l * NZP + tid] * rv[l];
                qacc += v;
            }
            // no second barrier: the next stage writes the OTHER row-block buffer, the other T, the other g~ and the other rv before its
            // barrier; this stage's buffers are rewritten two stages on, i.e. after every thread has passed the next barrier
            continue;
        }
        for (int t = tid; t < n * NZP; t += T_) {   // element (pr, c) of the row block: c = t mod 128, no division
            const int c = t & (NZP - 1), pr = t >> 7;
            if (c < wcols) {
                double v;
                if (c >= k * m) v = Bk[(c - k * m) * n + pr];
                else {
                    v = 0.0;
                    for (int l = 0; l < n; ++l) v += Ak[l * n + pr] * Gc[l * NZP + c];
                }
                Gn[t] = v;
            }
        }
        if (tid >= T_ - n) {  // the last n threads (for n <= 7 they have no share of the row block)
            const int pr = tid - (T_ - n);
            double v = cs[k * n + pr];
            for (int l = 0; l < n; ++l) v += Ak[l * n + pr] * gk[l];
            gn[pr] = v;
            rv2[(k & 1) * n + pr] = v + es[k * n + pr];
        }
        __syncthreads();
        { double* t = Gc; Gc = Gn; Gn = t; }
        { double* t = gk; gk = gn; gn = t; }
        const double* rv = rv2 + (k & 1) * n;
        for (int t = tid; t < n * NZP; t += T_) {
            const int c = t & (NZP - 1), pr = t >> 7;
            if (c < wcols) {
                double v = 0.0;
                for (int l = 0; l < n; ++l) v += Qk[l * n + pr] * Gc[l * NZP + c];
                T[t] = v;
            }
        }
        if (tid < nab) {  // this stage's (A, B) slot is free (last read before the barrier above): it takes stage k + 2
            if (k + 2 < N) AB[(k & 1) * nab + tid] = pre;
            if (k + 3 < N) pre = ab_load(k + 3);
        }
        __syncthreads();
        const int ni = (wcols + DX - 1) / DX;  // DX-column groups that hold non-zeros
        for (int l = 0; l < n; ++l) {
            double a[TS], b[TS];
#pragma unroll
            for (int i = 0; i < TS; ++i) {
                a[i] = i < ni ? Gc[l * NZP + tx + DX * i] : 0.0;
                b[i] = i < ni ? T[l * NZP + ty + DX * i] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < TS; ++i)
#pragma unroll
                for (int j = 0; j < TS; ++j) acc[i][j] += a[i] * b[j];
        }
        if (tid < wcols) {
            double v = 0.0;
            for (int l = 0; l < n; ++l) v += T[l * NZP + tid] * rv[l];
            qacc += v;
        }
        // no barrier here: before its first barrier the next stage writes only the other row-block buffer, the other g~ and the
        // other rv; T and the (A, B) slot are rewritten after that barrier, i.e. after every thread has finished the reads above
    }
    double* H = p.H + inst * (size_t)nz * nz;
#pragma unroll
    for (int j = 0; j < TS; ++j)
#pragma unroll
        for (int i = 0; i < TS; ++i) {
            const int r = tx + DX * i, c = ty + DX * j;
            if (r < nz && c < nz) {
                double v = 2.0 * acc[i][j];
                const int ir = r / m, ar = r % m, ic = c / m, ac = c % m;
                if (p.useR && ir == ic) v += 2.0 * Rs[ac * m + ar];
                if (p.useS) {
                    if (ir == ic) v += 2.0 * ((ir <= N - 2 ? 1 : 0) + (ir >= 1 ? 1 : 0)) * Ss[ac * m + ar];
                    else if (ir - ic == 1 || ic - ir == 1) v -= 2.0 * Ss[ac * m + ar];
                }
                H[(size_t)c * nz + r] = v;
            }
        }
    const double qv = 2.0 * qacc + ((p.qadd && tid < nz) ? p.qadd[inst * nz + tid] : 0.0);
    if (tid < nz) p.q[inst * nz + tid] = qv;
    if (p.sc_Hs) {
        if (tid == 0) p.sc_flag[inst] = 0;
        __threadfence_block();
        __syncthreads();   // (every LDS buffer is dead: the scaling's 128 doubles sit at the start)
        design_scale_body(nz, p.nzs, 0, H, nullptr, p.sc_d + inst * (size_t)p.nzs, p.sc_Hs + inst * (size_t)nz * p.nzs, nullptr,
                          p.sc_flag + inst, 0, smem);
        if (tid < nz) p.sc_fS[inst * nz + tid] = qv * smem[tid];
    }
}

// fS_i = d_i .* g: the constant part of the scaled gradient (g = 2 D'Sbar D u_ref, shared or per instance) for every instance
inline __global__ __launch_bounds__(256) void k_fs_scale(int batch, int nz, int nzs, const double* g, long g_stride, const double* d,
                                                  double* fS) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < (long)batch * nz; t += (long)gridDim.x * blockDim.x) {
        const long i = t / nz;
        const int r = (int)(t % nz);
        fS[t] = g[i * g_stride + r] * d[i * nzs + r];
    }
}

// Per-step re-linearisation pipeline: an instance whose design failed (non-positive diagonal / pivot: bFlag != 0) is reported through
// its solve status instead of a host-side error, so that the pipeline needs no host round trip.
inline __global__ __launch_bounds__(256) void k_flag_to_status(int batch, const int* flag, int32_t* status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < batch && flag[i] != 0) status[i] = 2;  // ALMPC_NON_FINITE: no usable solution for this instance
}

}  // namespace almpc
