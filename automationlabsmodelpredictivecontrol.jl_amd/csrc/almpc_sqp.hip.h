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

namespace almpc {

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
    double *x, *ex, *u, *eu;    // result buffers: the iterate after the update
};

// Before the QP: defects, state errors and the input part of the gradient, one workgroup per instance.
__global__ __launch_bounds__(256) void k_sqp_prepare(SqpParams p) {
    const int n = p.n, m = p.m, N = p.N, nz = p.nz;
    const size_t i = blockIdx.x;
    const double* xb = p.xbar + i * (size_t)(N + 1) * n;
    const double* ub = p.ubar + i * (size_t)nz;
    for (int t = threadIdx.x; t < N * n; t += blockDim.x) {
        const double xn = xb[n + t];  // stage k+1, component j  (t = k*n + j)
        p.c[i * (size_t)N * n + t] = p.fval[i * (size_t)N * n + t] - xn;
        p.ebar[i * (size_t)N * n + t] = xn - p.xref[n + t];
    }
    for (int t = threadIdx.x; t < nz; t += blockDim.x) {
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

// After the QP: dx_{k+1} = A_k dx_k + B_k v_k + c_k (dx_0 = 0), xbar += s dx, ubar += s v; one wave per instance.
// Lane j < n owns component j of dx; the previous dx is exchanged through LDS.
__global__ __launch_bounds__(64) void k_sqp_step(SqpParams p) {
    __shared__ double dxs[2][64];
    __shared__ int skip;
    const int n = p.n, m = p.m, N = p.N, nz = p.nz, lane = threadIdx.x;
    const size_t i = blockIdx.x;
    const double* v = p.v + i * (size_t)nz;
    double* xb = p.xbar + i * (size_t)(N + 1) * n;
    double* ub = p.ubar + i * (size_t)nz;
    // largest step / defect of this instance, and the finite check
    double vmax = 0.0, cmax = 0.0;
    int finite = 1;
    for (int t = lane; t < nz; t += 64) {
        const double a = fabs(v[t]);
        finite &= (a <= 1.79e308) ? 1 : 0;
        vmax = fmax(vmax, a);
    }
    for (int t = lane; t < N * n; t += 64) {
        const double a = fabs(p.c[i * (size_t)N * n + t]);
        finite &= (a <= 1.79e308) ? 1 : 0;
        cmax = fmax(cmax, a);
    }
    finite = __all(finite);
    if (lane == 0) {
        const int s = (!finite || p.flag[i] != 0 || p.status[i] == 2) ? 1 : 0;
        skip = s;
        if (s) p.bad[i] = 1;
    }
    __syncthreads();
    if (!skip) {
        vmax = wave_max(vmax);
        cmax = wave_max(cmax);
        if (lane == 0) {
            atomicMax(p.stats + 0, (unsigned long long)__double_as_longlong(vmax));
            atomicMax(p.stats + 1, (unsigned long long)__double_as_longlong(cmax));
        }
        if (lane < n) dxs[0][lane] = 0.0;
        __syncthreads();
        for (int k = 0; k < N; ++k) {
            const double* A = p.A + (i * N + k) * (size_t)n * n;
            const double* B = p.B + (i * N + k) * (size_t)n * m;
            const double* dxp = dxs[k & 1];
            if (lane < n) {
                double s = p.c[(i * N + k) * (size_t)n + lane];
                for (int c2 = 0; c2 < n; ++c2) s += A[(size_t)c2 * n + lane] * dxp[c2];
                for (int c2 = 0; c2 < m; ++c2) s += B[(size_t)c2 * n + lane] * v[k * m + c2];
                dxs[(k + 1) & 1][lane] = s;
                xb[(size_t)(k + 1) * n + lane] += p.step_scale * s;
            }
            __syncthreads();
        }
        for (int t = lane; t < nz; t += 64) {
            const int a = t % m;
            ub[t] = fmin(fmax(ub[t] + p.step_scale * v[t], p.umin[a]), p.umax[a]);  // v is feasible: the clip only removes rounding
        }
        __syncthreads();
    }
    // results: the iterate itself (also for a skipped instance: its last good iterate)
    for (int t = lane; t < (N + 1) * n; t += 64) {
        const double xv = xb[t];
        p.x[i * (size_t)(N + 1) * n + t] = xv;
        p.ex[i * (size_t)(N + 1) * n + t] = xv - p.xref[t];
    }
    for (int t = lane; t < nz; t += 64) {
        const double uv = ub[t];
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

__global__ __launch_bounds__(256) void k_fnn_rollout(FnnRolloutParams p) {
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
