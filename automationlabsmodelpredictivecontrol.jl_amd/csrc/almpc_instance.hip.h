// almpc_instance.hip.h -- per-instance models (almpc_design_batched): every instance has its own (A_i, B_i) and therefore its
// own condensed Hessian, scaling, KKT inverse and H'^-1.  This is the regime of BASELINE.json configs[3] (a black-box model
// re-linearised at every instance's current state): the reference's QP for given (A, B)
// (src/sub/model_modeler_implementation/linear/mpc_modeler_implementation_linear.jl:48-100, src/sub/design_mpc.jl:405-468) is
// the same, only nothing is shared across the batch any more, so there is no 16-instance MFMA tile with a common A operand.
//
//   design : the kernels of almpc_design.hip.h with blockIdx.y = instance (MFMA contraction Gamma' Qbar Gamma per instance)
//   k_admm_inst : ONE WORKGROUP PER INSTANCE.  The instance's KKT inverse M_i^-1 (nz x nzs doubles, 115 KB for nz = 120) is
//            copied to LDS once and all K iterations run out of LDS: HBM traffic per instance-step is one read of M_i^-1
//            (+ F'_i, V_i) instead of K reads.  The symmetric matrix is walked by columns (lane r reads M[c][r]: consecutive
//            lanes, consecutive addresses, no bank conflict), the right-hand side is a broadcast read.
//   polish / rollout : k_polish<false> with per-instance strides for G, d, A, B (almpc_kernels.hip.h).
// Iteration formulas, termination test and outputs are those of k_admm (OSQP Algorithm 1, box form).
#pragma once
#include "almpc_kernels.hip.h"

namespace almpc {

struct AdmmInstParams {
    int nz, n, m, batch, nzs;
    const double* Minv;   // [batch][nz][nzs]  (H'_i + sigma I + diag(rho_i))^-1, symmetric
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
    int32_t* status; int32_t* iters; int32_t* piters; int32_t* overflow; int32_t* perm;
    double sigma, alpha, eps_abs, eps_rel;
    int max_iter, check_every, warm;
};

constexpr int ADMM_INST_THREADS = 256;  // two column halves x 128 rows

__global__ __launch_bounds__(ADMM_INST_THREADS) void k_admm_inst(AdmmInstParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int nz = p.nz, nzs = p.nzs, n = p.n;
    const int inst = blockIdx.x;
    double* M = smem;                       // [nz][nzs]
    double* rhs0 = M + (size_t)nz * nzs;    // [nzs]
    double* rhs1 = rhs0 + nzs;              // [nzs]
    double* part = rhs1 + nzs;              // [2][nzs] partial products of the two column halves
    double* red = part + 2 * nzs;           // [4 waves][8]
    double* e0s = red + 32;                 // [n]
    const int tid = threadIdx.x, r = tid & 127, half = tid >> 7, wv = tid >> 6;
    const bool own = half == 0 && r < nz;   // this thread carries row r of the iterate

    // ---- M_i^-1 -> LDS (16-byte copies, 8 in flight per thread)
    {
        const d2* src = reinterpret_cast<const d2*>(p.Minv + (size_t)inst * nz * nzs);
        d2* dst = reinterpret_cast<d2*>(M);
        const int cnt2 = nz * nzs / 2;
        for (int i0 = tid; i0 < cnt2; i0 += ADMM_INST_THREADS * 8) {
            d2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * ADMM_INST_THREADS; v[u] = src[i < cnt2 ? i : 0]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * ADMM_INST_THREADS; if (i < cnt2) dst[i] = v[u]; }
        }
    }
    if (tid < n) e0s[tid] = p.x0[(size_t)inst * n + tid] - p.xref[(size_t)inst * p.xref_stride + tid];
    __syncthreads();

    // ---- per-row constants, f' = F' e0 + fS, v0 = V e0 + v0S
    double dv = 1.0, dinv = 1.0, lo = 0.0, hi = 0.0, fs = 0.0, v0 = 0.0, rho = 1.0;
    if (own) {
        dv = p.dvec[(size_t)inst * nzs + r];
        dinv = 1.0 / dv;
        rho = p.rhovec[(size_t)inst * nzs + r];
        const double ur = p.uref[(size_t)inst * p.uref_stride + r];
        lo = (p.umin[r % p.m] - ur) * dinv;
        hi = (p.umax[r % p.m] - ur) * dinv;
        const double* Fi = p.Fs + (size_t)inst * n * nzs;
        const double* Vi = p.Vs + (size_t)inst * n * nzs;
        double af = p.fS[(size_t)inst * nz + r], av = p.v0S[(size_t)inst * nz + r];
        for (int c = 0; c < n; ++c) {
            const double e = e0s[c];
            af += Fi[(size_t)c * nzs + r] * e;
            av += Vi[(size_t)c * nzs + r] * e;
        }
        fs = af; v0 = av;
    }
    // |f/d|_inf (constant part of the dual tolerance)
    auto block_max = [&](double v, int slot) -> double {
        v = wave_max(v);
        if ((tid & 63) == 0) red[wv * 8 + slot] = v;
        __syncthreads();
        const double o = fmax(fmax(red[0 * 8 + slot], red[1 * 8 + slot]), fmax(red[2 * 8 + slot], red[3 * 8 + slot]));
        __syncthreads();
        return o;
    };
    const double nf = block_max(own ? fabs(fs * dinv) : 0.0, 0);

    // ---- initial iterate (yt = y / rho_r)
    double x = 0.0, z = 0.0, yt = 0.0, px = 0.0;
    const double sigma = p.sigma, alpha = p.alpha;
    if (p.warm) {
        if (own) {
            const size_t o = (size_t)inst * nzs + r;
            x = p.xs[o];
            yt = p.ys[o] / rho;
            z = fmin(fmax(p.zs[o], lo), hi);
        }
        if (half == 0) rhs0[r] = own ? x : 0.0;
        __syncthreads();
        if (own) {  // px = H'_i x: one product with the instance's scaled Hessian, read by columns from global memory
            const double* Hi = p.Hs + (size_t)inst * nz * nzs;
            double a = 0.0;
            for (int c = 0; c < nz; ++c) a += Hi[(size_t)c * nzs + r] * rhs0[c];
            px = a;
        }
        __syncthreads();
    }
    double rown = sigma * x - fs + rho * (z - yt);
    if (half == 0 && r < nzs) rhs0[r] = own ? rown : 0.0;
    __syncthreads();

    bool active = true;
    int my_iters = p.max_iter, my_status = 1;
    double* cur = rhs0;
    double* nxt = rhs1;
    const int c_lo = half == 0 ? 0 : nz / 2, c_hi = half == 0 ? nz / 2 : nz;
    const int rr = r < nzs ? r : 0;
    for (int it = 1; it <= p.max_iter; ++it) {
        {   // this half's share of row r of M_i^-1 rhs (M symmetric: column walk)
            double a0 = 0.0, a1 = 0.0;
            int c = c_lo;
            for (; c + 1 < c_hi; c += 2) {
                a0 += M[(size_t)c * nzs + rr] * cur[c];
                a1 += M[(size_t)(c + 1) * nzs + rr] * cur[c + 1];
            }
            if (c < c_hi) a0 += M[(size_t)c * nzs + rr] * cur[c];
            if (r < nzs) part[half * nzs + r] = a0 + a1;
        }
        __syncthreads();
        if (own) {
            const double xt = part[r] + part[nzs + r];
            if (active) {
                const double hxt = rown - (sigma + rho) * xt;  // H' xt, from the KKT identity
                px = alpha * hxt + (1.0 - alpha) * px;
                x = alpha * xt + (1.0 - alpha) * x;
                const double w = alpha * xt + (1.0 - alpha) * z + yt;
                const double zn = fmin(fmax(w, lo), hi);
                yt = w - zn;
                z = zn;
                rown = sigma * x - fs + rho * (z - yt);
            }
            nxt[r] = rown;
        }
        const bool check = (it % p.check_every == 0) || (it == p.max_iter);
        if (check) {  // block-uniform branch
            const double yi = rho * yt;
            const double rp = block_max(own ? fabs(dv * (x - z)) : 0.0, 0);
            const double nx = block_max(own ? fabs(dv * x) : 0.0, 1);
            const double nzn = block_max(own ? fabs(dv * z) : 0.0, 2);
            const double rd = block_max(own ? fabs((px + fs + yi) * dinv) : 0.0, 3);
            const double nhx = block_max(own ? fabs(px * dinv) : 0.0, 4);
            const double ny = block_max(own ? fabs(yi * dinv) : 0.0, 5);
            const double s = x + yi + px;
            const double bad = block_max((own && !(fabs(s) <= 1.79e308)) ? 1.0 : 0.0, 6);
            if (active) {
                const bool conv = (rp <= p.eps_abs + p.eps_rel * fmax(nx, nzn)) &&
                                  (rd <= p.eps_abs + p.eps_rel * fmax(fmax(nhx, ny), nf));
                if (bad > 0.0) { active = false; my_iters = it; my_status = 2; }
                else if (conv) { active = false; my_iters = it; my_status = 0; }
            }
            if (!active) break;  // block-uniform: every thread saw the same reduced values
        } else {
            __syncthreads();
        }
        double* t = cur; cur = nxt; nxt = t;
    }

    // ---- hand-off to the polish: same arrays as k_admm
    if (tid == 0) {
        p.iters[inst] = my_iters;
        p.status[inst] = my_status;
        p.piters[inst] = 0;
        p.overflow[inst] = 0;
        p.perm[inst] = inst;  // no ranking across instances here: processing order = instance order
    }
    if (half == 0 && r < nzs) {
        const size_t o = (size_t)inst * nzs + r;
        p.xs[o] = own ? x : 0.0;
        p.zs[o] = own ? z : 0.0;
        p.ys[o] = own ? rho * yt : 0.0;
        p.v0[o] = own ? v0 : 0.0;
    }
}

// fS_i = d_i .* g: the constant part of the scaled gradient (g = 2 D'Sbar D u_ref, shared or per instance) for every instance
__global__ __launch_bounds__(256) void k_fs_scale(int batch, int nz, int nzs, const double* g, long g_stride, const double* d,
                                                  double* fS) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < (long)batch * nz; t += (long)gridDim.x * blockDim.x) {
        const long i = t / nz;
        const int r = (int)(t % nz);
        fS[t] = g[i * g_stride + r] * d[i * nzs + r];
    }
}

}  // namespace almpc
