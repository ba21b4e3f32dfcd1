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
//   k_polish<WL>    exact active-set finish, one wave per instance, (G_WW)^-1 in LDS.
//   k_rollout       u, e_u, x, e_x from w (recursive e+ = A e + B v).
//
// MFMA used: v_mfma_f64_16x16x4_f64.  Lane l holds A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
// C/D: col = l&15, row = (l>>4) + 4*reg  (cdna_hip_programming.md section 3, f64 map).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace almpc {

typedef double d4 __attribute__((ext_vector_type(4)));

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
    const double* GFrag;       // [NRB][KS][64]   H'^-1
    const double* HFrag;       // [NRB][KS][64]   H'   (warm start only)
    const double* FFrag;       // [NRB][KSF][64]  F' = D F
    int ksf;
    const double* dvec;        // [nzs] scaling d (pad rows: 1)
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
    double rho, sigma, alpha, eps_abs, eps_rel;
    int max_iter, check_every, warm;
};

// One workgroup = NRB waves = one tile of 16 instances.  Wave w owns rows 16w..16w+15 of every
// instance vector in the MFMA C/D layout: lane (q = l>>4, col = l&15) holds rows 16w + q + 4i,
// i = 0..3, of instance col -- so the whole ADMM vector update is register-local and only the
// right-hand side travels through LDS (double-buffered, one barrier per iteration).
template <int NRB, int KS>
__global__ __launch_bounds__(64 * NRB) void k_admm(AdmmParams p) {
    constexpr int RP = 16 * NRB;  // padded rows
    static_assert(4 * KS <= RP, "K padding must fit the row padding");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* rhs0 = smem;                    // [RP][16]
    double* rhs1 = smem + RP * TILE;        // [RP][16]
    double* red = smem + 2 * RP * TILE;     // [NRB][8][16]
    double* e0s = red + NRB * 8 * TILE;     // [4*ksf][16]

    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int col = lane & 15, q = lane >> 4;
    const int inst = blockIdx.x * TILE + col;
    const bool valid = inst < p.batch;
    const int instc = valid ? inst : p.batch - 1;  // clamp: pad columns recompute the last instance, never stored

    // ---- shared KKT inverse -> registers (A fragments), coalesced 512 B per wave-instruction
    double a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = p.MinvFrag[((size_t)(wv * KS + ks)) * 64 + lane];

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
    __syncthreads();

    // ---- per-row constants and f' = F' e0 + fS
    int row[4];
    double dv[4], dinv[4], lo[4], hi[4], fs[4];
    {
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int ks = 0; ks < p.ksf; ++ks) {
            double af = p.FFrag[((size_t)(wv * p.ksf + ks)) * 64 + lane];
            double b = e0s[(4 * ks + q) * TILE + col];
            acc = mfma_f64(af, b, acc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            row[i] = wv * 16 + q + 4 * i;
            const bool in = row[i] < p.nz;
            const int r = in ? row[i] : 0;
            dv[i] = in ? p.dvec[r] : 1.0;
            dinv[i] = 1.0 / dv[i];
            const double ur = p.uref[(size_t)instc * p.uref_stride + r];
            lo[i] = in ? (p.umin[r % p.m] - ur) * dinv[i] : 0.0;
            hi[i] = in ? (p.umax[r % p.m] - ur) * dinv[i] : 0.0;
            fs[i] = in ? acc[i] + p.fS[(size_t)instc * p.fS_stride + r] : 0.0;
        }
    }

    // ---- initial iterate
    double x[4], z[4], y[4], px[4], rown[4];
    const double rho = p.rho, sigma = p.sigma, alpha = p.alpha, rho_inv = 1.0 / p.rho;
    if (p.warm) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t o = (size_t)instc * p.nzs + row[i];
            x[i] = p.xs[o];
            y[i] = p.ys[o];
            z[i] = fmin(fmax(p.zs[o], lo[i]), hi[i]);
            rhs0[row[i] * TILE + col] = x[i];
        }
        __syncthreads();
        // px = H' x needs one true product: stream the H' fragments once
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int ks = 0; ks < KS; ++ks) {
            double ah = p.HFrag[((size_t)(wv * KS + ks)) * 64 + lane];
            acc = mfma_f64(ah, rhs0[(4 * ks + q) * TILE + col], acc);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) px[i] = acc[i];
        __syncthreads();
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = z[i] = y[i] = px[i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rown[i] = sigma * x[i] - fs[i] + rho * z[i] - y[i];
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
                const double hxt = rown[i] - (sigma + rho) * xt;  // H' xt, from the KKT identity
                px[i] = alpha * hxt + (1.0 - alpha) * px[i];
                x[i] = alpha * xt + (1.0 - alpha) * x[i];
                const double w = alpha * xt + (1.0 - alpha) * z[i] + y[i] * rho_inv;
                const double zn = fmin(fmax(w, lo[i]), hi[i]);
                y[i] = rho * (w - zn);
                z[i] = zn;
                rown[i] = sigma * x[i] - fs[i] + rho * z[i] - y[i];
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
                m_rd = fmax(m_rd, fabs((px[i] + fs[i] + y[i]) * dinv[i]));
                m_hx = fmax(m_hx, fabs(px[i] * dinv[i]));
                m_y = fmax(m_y, fabs(y[i] * dinv[i]));
                const double s = x[i] + y[i] + px[i];
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

    // ---- results of the ADMM stage (scaled coordinates)
    if (valid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t o = (size_t)inst * p.nzs + row[i];
            p.xs[o] = x[i];
            p.zs[o] = z[i];
            p.ys[o] = y[i];
        }
        if (wv == 0 && q == 0) {
            p.iters[inst] = my_iters;
            p.status[inst] = my_status;
        }
    }

    // ---- v0 = -H'^-1 f' for the polish: same tile product with the G fragments
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = p.GFrag[((size_t)(wv * KS + ks)) * 64 + lane];
    __syncthreads();  // everyone is done reading cur/nxt
#pragma unroll
    for (int i = 0; i < 4; ++i) nxt[row[i] * TILE + col] = -fs[i];
    __syncthreads();
    const d4 v04 = tile_matmul<KS>(a, nxt, q, col);
    if (valid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) p.v0[(size_t)inst * p.nzs + row[i]] = v04[i];
    }
}

// ------------------------------------------------------------------------------------------------
// polish: primal active-set finish with the shared inverse G = H'^-1 (oracle: polish_active_set)
// ------------------------------------------------------------------------------------------------
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
    const double* v0;
    double* w;           // result (scaled), [batch][nzs]
    int32_t* status;     // in: ADMM status; out: final
    int32_t* piters;
    int32_t* overflow;   // per instance: 1 if the working set outgrew WL (handled by the next tier)
    int max_iter;
    int tier;            // 0: process every finite instance; >0: only those flagged by the previous tier
};

__device__ __forceinline__ double readlane_d(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
// (min value, smallest index among ties) over the wave; every lane gets the result
__device__ __forceinline__ void wave_argmin(double& v, int& idx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(idx, o);
        if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

// One wave per instance.  nz <= 128: lane l holds rows l and l+64.  Working set size k <= WL <= 64:
// lane i < k owns row/column i of Sinv = (G_WW)^-1, stored column-major in LDS (S[c*WL + r]) so that a
// sweep over columns reads consecutive addresses across lanes.
template <int WL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_polish(PolishParams p) {
    static_assert(WL <= 64, "one lane per working-set row");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int PER_WAVE = WL * WL + WL + WL;  // S | Wb | (Widx, Wsd as int pairs)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int inst = blockIdx.x * WAVES + wv;
    if (inst >= p.batch) return;
    const int st_in = p.status[inst];
    if (st_in == 2) return;  // non-finite: nothing to polish
    if (p.tier > 0 && p.overflow[inst] != p.tier) return;

    double* S = smem + (size_t)wv * PER_WAVE;
    double* Wb = S + WL * WL;
    int* Widx = reinterpret_cast<int*>(Wb + WL);
    int* Wsd = Widx + WL;

    const int nz = p.nz, nzs = p.nzs;
    const size_t base = (size_t)inst * nzs;
    // rows of this lane
    const int r0 = lane, r1 = lane + 64;
    const bool in0 = r0 < nz, in1 = r1 < nz;
    double lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0, v00 = 0, v01 = 0, w0 = 0, w1 = 0, y0 = 0, y1 = 0;
    if (in0) {
        const double di = 1.0 / p.dvec[r0];
        const double ur = p.uref[(size_t)inst * p.uref_stride + r0];
        lo0 = (p.umin[r0 % p.m] - ur) * di; hi0 = (p.umax[r0 % p.m] - ur) * di;
        v00 = p.v0[base + r0]; y0 = p.ys[base + r0];
        w0 = fmin(fmax(p.zs[base + r0], lo0), hi0);
    }
    if (in1) {
        const double di = 1.0 / p.dvec[r1];
        const double ur = p.uref[(size_t)inst * p.uref_stride + r1];
        lo1 = (p.umin[r1 % p.m] - ur) * di; hi1 = (p.umax[r1 % p.m] - ur) * di;
        v01 = p.v0[base + r1]; y1 = p.ys[base + r1];
        w1 = fmin(fmax(p.zs[base + r1], lo1), hi1);
    }
    int pos0 = -1, pos1 = -1;  // position of the lane's rows in W (or -1)
    int k = 0;                 // |W| (wave-uniform)
    bool overflow = false;

    // add row j (uniform) at bound value bval with side sd (+1 upper, -1 lower)
    auto add_row = [&](int j, double bval, int sd) {
        // c = G[W, j]; u = Sinv c; sc = G_jj - c'u
        double c = 0.0;
        if (lane < k) c = p.G[(size_t)Widx[lane] * nzs + j];
        double u = 0.0;
        for (int l = 0; l < k; ++l) {
            const double cl = readlane_d(c, l);
            if (lane < k) u += S[l * WL + lane] * cl;
        }
        const double sc = p.G[(size_t)j * nzs + j] - wave_sum(lane < k ? c * u : 0.0);
        const double isc = 1.0 / sc;
        for (int l = 0; l < k; ++l) {
            const double ul = readlane_d(u, l);
            if (lane < k) S[l * WL + lane] += u * ul * isc;
        }
        if (lane < k) {
            S[k * WL + lane] = -u * isc;  // new column k
            S[lane * WL + k] = -u * isc;  // new row k
        }
        if (lane == 0) {
            S[k * WL + k] = isc;
            Widx[k] = j; Wsd[k] = sd; Wb[k] = bval;
        }
        if (j == r0) pos0 = k;
        if (j == r1) pos1 = k;
        k += 1;
        wave_lds_sync();
    };
    // remove position pos (uniform): Schur down-date, then move the last row/column into the hole
    auto remove_pos = [&](int pos) {
        double pc = 0.0;
        if (lane < k) pc = S[pos * WL + lane];
        const double ipp = 1.0 / readlane_d(pc, pos);
        for (int l = 0; l < k; ++l) {
            const double pl = readlane_d(pc, l);
            if (lane < k) S[l * WL + lane] -= pc * pl * ipp;
        }
        wave_lds_sync();
        const int last = k - 1;
        const int jrem = Widx[pos], jlast = Widx[last];
        if (pos != last) {
            double colv = 0.0;
            if (lane < k) colv = S[last * WL + lane];  // column `last` (== row `last`, symmetric)
            const double corner = readlane_d(colv, last);
            wave_lds_sync();
            if (lane < k && lane != pos && lane != last) {
                S[pos * WL + lane] = colv;
                S[lane * WL + pos] = colv;
            }
            if (lane == 0) {
                S[pos * WL + pos] = corner;
                Widx[pos] = jlast; Wsd[pos] = Wsd[last]; Wb[pos] = Wb[last];
            }
        }
        if (r0 == jrem) pos0 = -1;
        if (r1 == jrem) pos1 = -1;
        if (pos != last) {
            if (r0 == jlast) pos0 = pos;
            if (r1 == jlast) pos1 = pos;
        }
        k -= 1;
        wave_lds_sync();
    };

    // ---- initial working set from the ADMM multipliers (OSQP polish rule: sign of y), rows in order
    for (int j = 0; j < nz && !overflow; ++j) {
        const int src = j & 63;
        const bool second = j >= 64;
        const double yy = readlane_d(second ? y1 : y0, src);
        const double ww = readlane_d(second ? w1 : w0, src);
        const double ll = readlane_d(second ? lo1 : lo0, src);
        const double hh = readlane_d(second ? hi1 : hi0, src);
        int sd = 0;
        if (yy < 0.0 && ww <= ll) sd = -1;
        else if (yy > 0.0 && ww >= hh) sd = +1;
        if (sd != 0) {
            if (k == WL) { overflow = true; break; }
            add_row(j, sd > 0 ? hh : ll, sd);
        }
    }

    int it = 0;
    int fin = 1;  // 0 = certified
    const int max_iter = p.max_iter;
    while (!overflow && it < max_iter) {
        ++it;
        // ---- face minimiser t = v0 - G[:,W] lam,  lam = Sinv (v0_W - b)
        double lam = 0.0;
        if (k > 0) {
            // r_i = v0[W_i] - b_i : lane i fetches v0 of row Widx[i] from its owner lane
            const int jw = (lane < k) ? Widx[lane] : 0;
            const double a0 = __shfl(v00, jw & 63);
            const double a1 = __shfl(v01, jw & 63);
            const double r = (lane < k) ? ((jw >= 64 ? a1 : a0) - Wb[lane]) : 0.0;
            for (int l = 0; l < k; ++l) {
                const double rl = readlane_d(r, l);
                if (lane < k) lam += S[l * WL + lane] * rl;
            }
        }
        double t0 = v00, t1 = v01;
        for (int l = 0; l < k; ++l) {
            const int j = Widx[l];
            const double ll = readlane_d(lam, l);
            if (in0) t0 -= p.G[(size_t)j * nzs + r0] * ll;
            if (in1) t1 -= p.G[(size_t)j * nzs + r1] * ll;
        }
        if (pos0 >= 0) t0 = Wb[pos0];
        if (pos1 >= 0) t1 = Wb[pos1];
        // ---- ratio test over the free rows
        double rr = __builtin_inf();
        int rj = 0x7fffffff;
        int rside = 0;
        if (in0 && pos0 < 0) {
            if (t0 > hi0) { rr = (hi0 - w0) / (t0 - w0); rj = r0; rside = +1; }
            else if (t0 < lo0) { rr = (lo0 - w0) / (t0 - w0); rj = r0; rside = -1; }
        }
        if (in1 && pos1 < 0) {
            double c = __builtin_inf();
            int s = 0;
            if (t1 > hi1) { c = (hi1 - w1) / (t1 - w1); s = +1; }
            else if (t1 < lo1) { c = (lo1 - w1) / (t1 - w1); s = -1; }
            if (c < rr) { rr = c; rj = r1; rside = s; }  // r1 > r0: ties keep the smaller row
        }
        double rmin = rr;
        int jmin = rj;
        wave_argmin(rmin, jmin);
        jmin = __builtin_amdgcn_readfirstlane(jmin);
        if (rmin < 1.0) {
            const double tt = fmax(rmin, 0.0);
            if (in0 && pos0 < 0) w0 += tt * (t0 - w0);
            if (in1 && pos1 < 0) w1 += tt * (t1 - w1);
            // owner lane of jmin knows the side and the bound
            const int owner = jmin & 63;
            int sd = (rj == jmin) ? rside : 0;
            sd = __shfl(sd, owner);
            const double bh = __shfl(jmin >= 64 ? hi1 : hi0, owner);
            const double bl = __shfl(jmin >= 64 ? lo1 : lo0, owner);
            const double bval = sd > 0 ? bh : bl;
            if (r0 == jmin) w0 = bval;
            if (r1 == jmin) w1 = bval;
            if (k == WL) { overflow = true; break; }
            add_row(jmin, bval, sd);
            continue;
        }
        w0 = t0; w1 = t1;
        if (k == 0) { fin = 0; break; }
        // ---- multiplier signs: upper bound needs lam >= 0, lower bound lam <= 0
        double viol = -__builtin_inf();
        int vi = 0x7fffffff;
        if (lane < k) { viol = (Wsd[lane] > 0) ? -lam : lam; vi = lane; }
        const double lmax = wave_max(lane < k ? fabs(lam) : 0.0);
        double nv = -viol;
        wave_argmin(nv, vi);  // argmax of viol, smallest position among ties
        vi = __builtin_amdgcn_readfirstlane(vi);
        if (-nv <= 1e-12 * fmax(1.0, lmax)) { fin = 0; break; }
        remove_pos(vi);
    }

    if (overflow) {
        if (lane == 0) p.overflow[inst] = p.tier + 1;
        return;
    }
    if (in0) p.w[base + r0] = fmin(fmax(w0, lo0), hi0);
    if (in1) p.w[base + r1] = fmin(fmax(w1, lo1), hi1);
    if (lane == 0) {
        p.piters[inst] = it;
        p.status[inst] = (fin == 0) ? 0 : st_in;
        p.overflow[inst] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// rollout: outputs of calculate! (src/main/computation_mpc.jl:50-53)
//   e_u = reshape(d .* w), u = e_u + u_ref, e_x[:,1] = x0 - x_ref[:,1], e_x[:,k+1] = A e_x[:,k] + B e_u[:,k],
//   x = e_x + x_ref.   One wave per instance, lane i < n owns state row i.
// ------------------------------------------------------------------------------------------------
struct RolloutParams {
    int n, m, N, batch, nzs;
    const double* A;  // n*n column-major
    const double* B;  // n*m column-major
    const double* dvec;
    const double* w;  // scaled solution [batch][nzs]
    const double* x0;
    const double* xref;  // [xref_stride*inst + n*k + i]
    long xref_stride;
    const double* uref;
    long uref_stride;
    double* x;   // [batch][N+1][n]
    double* ex;
    double* u;   // [batch][N][m]
    double* eu;
};

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_rollout(RolloutParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n = p.n, m = p.m, N = p.N;
    double* As = smem;                 // [n][n]  As[j*n + i] = A[i][j] (column-major as given)
    double* Bs = As + n * n;           // [m][n]
    double* ebuf = Bs + n * m;         // per wave: e (n) + v (m*N)
    for (int t = threadIdx.x; t < n * n; t += blockDim.x) As[t] = p.A[t];
    for (int t = threadIdx.x; t < n * m; t += blockDim.x) Bs[t] = p.B[t];
    __syncthreads();
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int inst = blockIdx.x * WAVES + wv;
    if (inst >= p.batch) return;
    const int nz = m * N;
    double* e = ebuf + (size_t)wv * (n + nz);
    double* v = e + n;
    // inputs: u, e_u
    for (int r = lane; r < nz; r += 64) {
        const double ev = p.w[(size_t)inst * p.nzs + r] * p.dvec[r];
        v[r] = ev;
        p.eu[(size_t)inst * nz + r] = ev;
        p.u[(size_t)inst * nz + r] = ev + p.uref[(size_t)inst * p.uref_stride + r];
    }
    const size_t xo = (size_t)inst * n * (N + 1);
    double ei = 0.0;
    if (lane < n) {
        ei = p.x0[(size_t)inst * n + lane] - p.xref[(size_t)inst * p.xref_stride + lane];
        e[lane] = ei;
        p.ex[xo + lane] = ei;
        p.x[xo + lane] = p.x0[(size_t)inst * n + lane];
    }
    wave_lds_sync();
    for (int k = 0; k < N; ++k) {
        double acc = 0.0;
        if (lane < n) {
            for (int j = 0; j < n; ++j) acc += As[j * n + lane] * e[j];
            for (int j = 0; j < m; ++j) acc += Bs[j * n + lane] * v[k * m + j];
        }
        wave_lds_sync();
        if (lane < n) {
            e[lane] = acc;
            p.ex[xo + (size_t)(k + 1) * n + lane] = acc;
            p.x[xo + (size_t)(k + 1) * n + lane] = acc + p.xref[(size_t)inst * p.xref_stride + (size_t)(k + 1) * n + lane];
        }
        wave_lds_sync();
    }
}

}  // namespace almpc
