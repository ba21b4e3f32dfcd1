// almpc_sqp.hip.h -- device side of the SQP outer loop for black-box (Fnn) models.
//
// The reference's NonLinearProgramming branch poses   min  e_x[:,N+1]' P e_x[:,N+1] + sum_k e_x[:,k]' Q e_x[:,k] + e_u[:,k]' R e_u[:,k]
// s.t.  x[:,k+1] = fnn(x[:,k], u[:,k]),  umin <= u <= umax   (.../fnn/mpc_modeler_implementation_fnn.jl:110-189, cost
// src/sub/design_mpc.jl:405-468) and hands it to Ipopt.  Here the same NLP is solved by Gauss-Newton SQP with multiple shooting:
// every outer iteration linearises the network along the current trajectory (k_fnn_jacobian), builds the time-varying condensed
// QP (k_design_ltv) and solves it with the per-instance step kernels.  The kernels below are the glue that keeps the whole
// iteration on the device: defects and gradient pieces before the QP, the trajectory update after it.
// The iterate (xbar, ubar) lives in the handle's per-instance reference buffers (dXref, dUref): the QP variable is v = u - ubar.
#pragma once
#include <hip/hip_runtime.h>
#include "almpc_fnn.hip.h"
#include "almpc_kernels.hip.h"  // wave_max
#include "almpc_instance.hip.h" // SqpParams, sqp_prepare_body

namespace almpc {

// (struct SqpParams and sqp_prepare_body: csrc/almpc_instance.hip.h, beside the design kernel whose head runs them)

// Before the QP: defects, state errors and the input part of the gradient, one workgroup per instance.
// (see sqp_prepare_body)
inline __global__ __launch_bounds__(256) void k_sqp_prepare(SqpParams p) { sqp_prepare_body(p, blockIdx.x); }

// After the QP: dx_{k+1} = A_k dx_k + B_k v_k + c_k (dx_0 = 0), xbar += s dx, ubar += s v; one workgroup per instance.
// The recursion is a dependent chain of N small products.  All of its operands are staged in LDS first -- [A_k | B_k | c_k] of
// `chunk` stages at a time with the whole workgroup (one HBM round trip per chunk; the host sizes the chunk, normally all N
// stages), v once -- and wave 0 then walks the stages on its own: lane j < n owns dx_j, the previous dx comes from the lanes
// themselves (v_readlane), so a stage is n + m FMAs deep with no barrier.  Trajectory update, results and the batch maxima
// are done by all threads around it.
__host__ __device__ inline int sqp_step_chunk(int n, int m, int N) {
    const long E = (long)n * n + (long)n * m + n, fixed = (long)(N + 1) * n + (long)m * N + 16;
    long ch = (12288 - fixed) / E;  // 96 KB of doubles
    return (int)(ch < 1 ? 1 : (ch > N ? N : ch));
}
__host__ __device__ inline size_t sqp_step_lds_doubles(int n, int m, int N) {
    return (size_t)sqp_step_chunk(n, m, N) * ((size_t)n * n + (size_t)n * m + n) + (size_t)(N + 1) * n + (size_t)m * N + 16;
}

inline __global__ __launch_bounds__(256) void k_sqp_step(SqpParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int skip;
    const int n = p.n, m = p.m, N = p.N, nz = p.nz, tid = threadIdx.x, nn = n * n, nm = n * m, E = nn + nm + n;
    const int CH = sqp_step_chunk(n, m, N);
    double* stage = smem;                       // [CH][E]
    double* dxa = stage + (size_t)CH * E;       // [(N+1)][n]: dx_0 = 0, dx_1, ...
    double* vs = dxa + (size_t)(N + 1) * n;     // [nz]
    double* red = vs + nz;                      // [16]
    const size_t i = blockIdx.x;
    const double* v = p.v + i * (size_t)nz;
    const double* cg = p.c + i * (size_t)N * n;
    double* xb = p.xbar + i * (size_t)(N + 1) * n;
    double* ub = p.ubar + i * (size_t)nz;
    const double* Ag = p.A + i * N * (size_t)nn;
    const double* Bg = p.B + i * N * (size_t)nm;
    // largest step / defect of this instance, and the finite check
    double vmax = 0.0, cmax = 0.0;
    int finite = 1;
    for (int t = tid; t < nz; t += 256) {
        const double a = v[t];
        vs[t] = a;
        finite &= (fabs(a) <= 1.79e308) ? 1 : 0;
        vmax = fmax(vmax, fabs(a));
    }
    for (int t = tid; t < N * n; t += 256) {
        const double a = fabs(cg[t]);
        finite &= (a <= 1.79e308) ? 1 : 0;
        cmax = fmax(cmax, a);
    }
    if (tid < n) dxa[tid] = 0.0;
    vmax = wave_max(vmax);
    cmax = wave_max(cmax);
    finite = __all(finite);
    if ((tid & 63) == 0) {
        red[tid >> 6] = vmax; red[4 + (tid >> 6)] = cmax; red[8 + (tid >> 6)] = finite ? 0.0 : 1.0;
    }
    __syncthreads();
    if (tid == 0) {
        const int redo_ = p.adaptive && p.mer[4 * i + 2] != 0.0;   // trial point rejected in k_sqp_prepare: this QP is void
        const int bad = !redo_ && ((red[8] + red[9] + red[10] + red[11] != 0.0) || p.flag[i] != 0 || p.status[i] == 2 || p.status[i] == 3);
        skip = bad || redo_;
        if (bad) p.bad[i] = 1;
        else if (!redo_) {
            atomicMax(p.stats + 0, (unsigned long long)__double_as_longlong(fmax(fmax(red[0], red[1]), fmax(red[2], red[3]))));
            atomicMax(p.stats + 1, (unsigned long long)__double_as_longlong(fmax(fmax(red[4], red[5]), fmax(red[6], red[7]))));
        }
    }
    __syncthreads();
    // Step length: `step_scale`, times the instance's own factor under step rule 1 (k_sqp_prepare).  An instance whose last
    // trial point was just rejected there sits on a new, shorter trial: its QP of this iteration is void.
    double scale = p.step_scale;
    if (p.adaptive) scale *= p.mer[4 * i];  // the factor in force for the step taken now (k_sqp_prepare has just updated it)
    if (!skip) {
        double dxr = 0.0;                       // wave 0, lane j < n: dx_j of the current stage
        const int lj = tid < n ? tid : 0;
        for (int k0 = 0; k0 < N; k0 += CH) {
            const int cnt = (N - k0 < CH) ? N - k0 : CH;
            if (k0 > 0) __syncthreads();        // the previous chunk has been consumed
            for (int t = tid; t < cnt * E; t += 256) {
                const int k = k0 + t / E, e = t % E;
                stage[t] = e < nn ? Ag[(size_t)k * nn + e] : (e < nn + nm ? Bg[(size_t)k * nm + e - nn] : cg[(size_t)k * n + e - nn - nm]);
            }
            __syncthreads();
            if (tid < 64) {
                for (int kk = 0; kk < cnt; ++kk) {
                    const double* A = stage + (size_t)kk * E;
                    const double* B = A + nn;
                    double s = A[nn + nm + lj];
                    for (int c2 = 0; c2 < n; ++c2) s += A[c2 * n + lj] * readlane_d(dxr, c2);
                    for (int c2 = 0; c2 < m; ++c2) s += B[c2 * n + lj] * vs[(k0 + kk) * m + c2];
                    dxr = s;
                    if (tid < n) dxa[(size_t)(k0 + kk + 1) * n + tid] = s;
                }
            }
        }
        __syncthreads();
    }
    // trajectory update and results: the iterate itself (for a skipped instance: its last good iterate)
    if (p.adaptive && !skip) {  // remember the point this step leaves and the (full) step itself
        for (int t = tid; t < (N + 1) * n; t += 256) {
            p.xback[i * (size_t)(N + 1) * n + t] = xb[t];
            p.dxback[i * (size_t)(N + 1) * n + t] = p.step_scale * dxa[t];
        }
        for (int t = tid; t < nz; t += 256) {
            p.uback[i * (size_t)nz + t] = ub[t];
            p.vback[i * (size_t)nz + t] = p.step_scale * vs[t];
        }
    }
    for (int t = tid; t < (N + 1) * n; t += 256) {
        double xv = xb[t];
        if (!skip && t >= n) { xv += scale * dxa[t]; xb[t] = xv; }
        p.x[i * (size_t)(N + 1) * n + t] = xv;
        p.ex[i * (size_t)(N + 1) * n + t] = xv - p.xref[t];
    }
    for (int t = tid; t < nz; t += 256) {
        double uv = ub[t];
        if (!skip) {
            const int a = t % m;
            uv = fmin(fmax(uv + scale * vs[t], p.umin[a]), p.umax[a]);  // v is feasible: the clip only removes rounding
            ub[t] = uv;
        }
        p.u[i * (size_t)nz + t] = uv;
        p.eu[i * (size_t)nz + t] = uv - p.uref[t];
    }
}

// Single-shooting start: xbar_0 = x0, xbar_{k+1} = fnn(xbar_k, ubar_k).  One workgroup per instance, thread i owns neuron i.
struct FnnRolloutParams {
    int n, m, H, L, act, N;
    const double* W_in; const double* W_h; const double* b_h; const double* W_out;
    const double* x0;     // [batch][n]
    const double* ubar;   // [batch][N][m]
    double* xbar;         // [batch][(N+1)][n]
};

inline __global__ __launch_bounds__(256) void k_fnn_rollout(FnnRolloutParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n = p.n, m = p.m, H = p.H, nin = n + m;
    double* y = smem;        // [H]
    double* yn = y + H;      // [H]
    double* z = yn + H;      // [nin]
    const size_t inst = blockIdx.x;
    double* xb = p.xbar + inst * (size_t)(p.N + 1) * n;
    const double* ub = p.ubar + inst * (size_t)p.N * m;
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const double v = p.x0[inst * n + t];
        z[t] = v;
        xb[t] = v;
    }
    __syncthreads();
    for (int k = 0; k < p.N; ++k) {
        for (int t = threadIdx.x; t < m; t += blockDim.x) z[n + t] = ub[k * m + t];
        __syncthreads();
        for (int i = threadIdx.x; i < H; i += blockDim.x) {
            double s = 0.0;
            for (int c = 0; c < nin; ++c) s += p.W_in[(size_t)c * H + i] * z[c];
            y[i] = s;
        }
        __syncthreads();
        for (int l = 0; l < p.L; ++l) {
            const double* W = p.W_h + (size_t)l * H * H;
            const double* b = p.b_h + (size_t)l * H;
            for (int i = threadIdx.x; i < H; i += blockDim.x) {
                double s = b[i];
                for (int j = 0; j < H; ++j) s += W[(size_t)j * H + i] * y[j];
                double val, der;
                fnn_act(p.act, s, val, der);
                yn[i] = val;
            }
            __syncthreads();
            for (int i = threadIdx.x; i < H; i += blockDim.x) y[i] = yn[i];
            __syncthreads();
        }
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            double s = 0.0;
            for (int j = 0; j < H; ++j) s += p.W_out[(size_t)j * n + i] * y[j];
            z[i] = s;
            xb[(size_t)(k + 1) * n + i] = s;
        }
        __syncthreads();
    }
}

}  // namespace almpc
