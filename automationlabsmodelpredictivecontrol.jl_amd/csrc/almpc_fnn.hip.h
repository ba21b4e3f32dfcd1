// almpc_fnn.hip.h -- batched linearisation of a black-box Fnn model (forward value + Jacobian per instance).
//
// Stands in for AutomationLabsSystems.proceed_system_linearization(system, x, u) (third-party, not in /root/reference;
// call sites: src/sub/model_modeler_implementation/fnn/mpc_modeler_implementation_fnn.jl:42-46,
// src/sub/design_mpc.jl:319-326), for the Fnn layout the reference reads from Flux.params (.../fnn/...:88-107, 127-144):
//     y1 = W_in [x;u] (no bias, no activation);  yj = act(W_h[j-2] y(j-1) + b_h[j-2]);  x+ = W_out y(L+1) (no bias).
// One workgroup (256 threads) per linearisation point; the H x (n+m) forward-mode Jacobian lives in LDS.
#pragma once
#include <hip/hip_runtime.h>

namespace almpc {

struct FnnParams {
    int n, m, H, L, act;  // act: 0 identity, 1 relu, 2 tanh, 3 sigmoid, 4 swish (x*sigmoid(x)): NNlib's names
    const double* W_in;   // H x (n+m) column-major
    const double* W_h;    // [L] H x H column-major
    const double* b_h;    // [L] H
    const double* W_out;  // n x H column-major
    int batch;
    const double* x;      // point t = (g, k), g = t / ppi, k = t % ppi:  x + g * xs_group + k * n   (ppi = 1, xs_group = n: [batch][n])
    const double* u;      //                                               u + g * us_group + k * m
    int ppi; long xs_group, us_group;  // points per group and group strides: the stages of one trajectory [ (N+1) n | N m ]
    double* A;            // [batch] n x n column-major
    double* B;            // [batch] n x m column-major
    double* f;            // [batch][n] or null
};

// activation value and derivative at pre-activation a
__device__ __forceinline__ void fnn_act(int act, double a, double& val, double& der) {
    switch (act) {
        case 1: val = fmax(a, 0.0); der = a > 0.0 ? 1.0 : 0.0; break;
        case 2: { const double t = tanh(a); val = t; der = 1.0 - t * t; break; }
        case 3: { const double s = 1.0 / (1.0 + exp(-a)); val = s; der = s * (1.0 - s); break; }
        case 4: { const double s = 1.0 / (1.0 + exp(-a)); val = a * s; der = s * (1.0 + a * (1.0 - s)); break; }
        default: val = a; der = 1.0; break;
    }
}

inline __global__ __launch_bounds__(256) void k_fnn_jacobian(FnnParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int n = p.n, m = p.m, H = p.H, nin = n + m;
    double* y = smem;             // [H]
    double* yn = y + H;           // [H]
    double* J = yn + H;           // [H][nin]  (row-major: J[i*nin + c])
    double* Jn = J + (size_t)H * nin;
    double* z = Jn + (size_t)H * nin;  // [nin]
    const int inst = blockIdx.x;
    const double* xp = p.x + (size_t)(inst / p.ppi) * p.xs_group + (size_t)(inst % p.ppi) * n;
    const double* up = p.u + (size_t)(inst / p.ppi) * p.us_group + (size_t)(inst % p.ppi) * m;
    for (int t = threadIdx.x; t < nin; t += blockDim.x) z[t] = t < n ? xp[t] : up[t - n];
    __syncthreads();
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        double s = 0.0;
        for (int c = 0; c < nin; ++c) s += p.W_in[(size_t)c * H + i] * z[c];
        y[i] = s;
    }
    for (int t = threadIdx.x; t < H * nin; t += blockDim.x) J[t] = p.W_in[(size_t)(t % nin) * H + t / nin];
    __syncthreads();
    for (int l = 0; l < p.L; ++l) {
        const double* W = p.W_h + (size_t)l * H * H;
        const double* b = p.b_h + (size_t)l * H;
        for (int i = threadIdx.x; i < H; i += blockDim.x) {
            double s = b[i];
            for (int j = 0; j < H; ++j) s += W[(size_t)j * H + i] * y[j];
            yn[i] = s;  // pre-activation
        }
        __syncthreads();
        for (int t = threadIdx.x; t < H * nin; t += blockDim.x) {
            const int i = t / nin, c = t % nin;
            double s = 0.0;
            for (int j = 0; j < H; ++j) s += W[(size_t)j * H + i] * J[(size_t)j * nin + c];
            double val, der;
            fnn_act(p.act, yn[i], val, der);
            Jn[t] = der == 0.0 ? 0.0 : der * s;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < H; i += blockDim.x) {
            double val, der;
            fnn_act(p.act, yn[i], val, der);
            y[i] = val;
        }
        for (int t = threadIdx.x; t < H * nin; t += blockDim.x) J[t] = Jn[t];
        __syncthreads();
    }
    for (int t = threadIdx.x; t < n * nin; t += blockDim.x) {
        const int i = t % n, c = t / n;
        double s = 0.0;
        for (int j = 0; j < H; ++j) s += p.W_out[(size_t)j * n + i] * J[(size_t)j * nin + c];
        if (c < n) p.A[(size_t)inst * n * n + (size_t)c * n + i] = s;
        else p.B[(size_t)inst * n * m + (size_t)(c - n) * n + i] = s;
    }
    if (p.f)
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            double s = 0.0;
            for (int j = 0; j < H; ++j) s += p.W_out[(size_t)j * n + i] * y[j];
            p.f[(size_t)inst * n + i] = s;
        }
}

// Wave-per-point build for small networks (the SQP loop linearises batch x N points per iteration): the weights are staged in
// LDS once per workgroup and shared by its 4 waves, every wave walks its own points with wave-local LDS buffers and no
// workgroup barrier inside the loop.  Same arithmetic, same summation order as k_fnn_jacobian (bit-identical results).
constexpr int FNN_W_WAVES = 4;

// (ppw: points per wave and pass, 1 or 2: see fnn_jacobian_point)
__host__ __device__ inline size_t fnn_w_lds_doubles(int n, int m, int H, int L, int ppw = 1) {
    const size_t nin = (size_t)n + m;
    const size_t weights = (size_t)H * nin + (size_t)L * H * H + (size_t)L * H + (size_t)n * H;
    const size_t per_wave = 2 * (size_t)H + 2 * (size_t)H * nin + nin;
    return weights + FNN_W_WAVES * ppw * per_wave;
}

// weights of the network into LDS (all threads of the workgroup; the caller synchronises): Win | Wh | bh | Wout
__device__ __forceinline__ void fnn_stage_weights(const FnnParams& p, double* smem) {
    const int n = p.n, m = p.m, H = p.H, L = p.L, nin = n + m;
    double* Win = smem;
    double* Wh = Win + (size_t)H * nin;
    double* bh = Wh + (size_t)L * H * H;
    double* Wout = bh + (size_t)L * H;
    for (int t = threadIdx.x; t < H * nin; t += blockDim.x) Win[t] = p.W_in[t];
    for (int t = threadIdx.x; t < L * H * H; t += blockDim.x) Wh[t] = p.W_h[t];
    for (int t = threadIdx.x; t < L * H; t += blockDim.x) bh[t] = p.b_h[t];
    for (int t = threadIdx.x; t < n * H; t += blockDim.x) Wout[t] = p.W_out[t];
}
__host__ __device__ inline size_t fnn_weights_doubles(int n, int m, int H, int L) {
    return (size_t)H * (n + m) + (size_t)L * H * H + (size_t)L * H + (size_t)n * H;
}
__host__ __device__ inline size_t fnn_wave_scratch_doubles(int n, int m, int H) { return 2 * (size_t)H + 2 * (size_t)H * (n + m) + (n + m); }

// Jacobians (and value) of the network at ONE point by ONE wave: weights staged at `wsm` (fnn_stage_weights), `y` = the wave's own
// scratch (fnn_wave_scratch_doubles).  Also the tail of nothing and the head of k_design_instance_t in the re-linearisation
// pipeline (the instance's workgroup linearises its own model: one launch less per step).
// LW: lanes per point (64: the wave works on one point; 32: each half-wave on its own -- small networks leave most lanes of a wave
// idle, and a point is a latency chain: two at a time halve the passes).  inst_ may lie beyond the batch (its half then computes on
// point 0 and stores nothing); lane: the lane within the point's LW lanes; y: the point's own scratch.
template <int LW = 64>
__device__ __forceinline__ void fnn_jacobian_point(const FnnParams& p, int inst_, int lane, const double* wsm, double* y) {
    const bool valid = inst_ < p.batch;
    const int inst = valid ? inst_ : 0;
    const int n = p.n, m = p.m, H = p.H, L = p.L, nin = n + m;
    const double* Win = wsm;                          // H x nin column-major
    const double* Wh = Win + (size_t)H * nin;         // [L] H x H column-major
    const double* bh = Wh + (size_t)L * H * H;        // [L] H
    const double* Wout = bh + (size_t)L * H;          // n x H column-major
    double* yn = y + H;
    double* J = yn + H;                               // [H][nin] row-major
    double* Jn = J + (size_t)H * nin;
    double* z = Jn + (size_t)H * nin;
    auto wsync = []() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    {
        const double* xp = p.x + (size_t)(inst / p.ppi) * p.xs_group + (size_t)(inst % p.ppi) * n;
        const double* up = p.u + (size_t)(inst / p.ppi) * p.us_group + (size_t)(inst % p.ppi) * m;
        for (int t = lane; t < nin; t += LW) z[t] = t < n ? xp[t] : up[t - n];
        wsync();
        for (int i = lane; i < H; i += LW) {
            double s = 0.0;
            for (int c = 0; c < nin; ++c) s += Win[(size_t)c * H + i] * z[c];
            y[i] = s;
        }
        for (int t = lane; t < H * nin; t += LW) J[t] = Win[(size_t)(t % nin) * H + t / nin];
        wsync();
        for (int l = 0; l < L; ++l) {
            const double* W = Wh + (size_t)l * H * H;
            const double* b = bh + (size_t)l * H;
            for (int i = lane; i < H; i += LW) {
                double s = b[i];
                for (int j = 0; j < H; ++j) s += W[(size_t)j * H + i] * y[j];
                yn[i] = s;  // pre-activation
            }
            wsync();
            for (int t = lane; t < H * nin; t += LW) {
                const int i = t / nin, c = t % nin;
                double s = 0.0;
                for (int j = 0; j < H; ++j) s += W[(size_t)j * H + i] * J[(size_t)j * nin + c];
                double val, der;
                fnn_act(p.act, yn[i], val, der);
                Jn[t] = der == 0.0 ? 0.0 : der * s;
            }
            wsync();
            for (int i = lane; i < H; i += LW) {
                double val, der;
                fnn_act(p.act, yn[i], val, der);
                y[i] = val;
            }
            for (int t = lane; t < H * nin; t += LW) J[t] = Jn[t];
            wsync();
        }
        for (int t = lane; t < n * nin; t += LW) {
            const int i = t % n, c = t / n;
            double s = 0.0;
            for (int j = 0; j < H; ++j) s += Wout[(size_t)j * n + i] * J[(size_t)j * nin + c];
            if (!valid) continue;
            if (c < n) p.A[(size_t)inst * n * n + (size_t)c * n + i] = s;
            else p.B[(size_t)inst * n * m + (size_t)(c - n) * n + i] = s;
        }
        if (p.f)
            for (int i = lane; i < n; i += LW) {
                double s = 0.0;
                for (int j = 0; j < H; ++j) s += Wout[(size_t)j * n + i] * y[j];
                if (valid) p.f[(size_t)inst * n + i] = s;
            }
        wsync();  // the next point overwrites z, y, J
    }
}

template <int LW>
__global__ __launch_bounds__(64 * FNN_W_WAVES) void k_fnn_jacobian_w(FnnParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int PPW = 64 / LW;   // points per wave and pass
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane / LW, hl = lane % LW;
    double* y = smem + fnn_weights_doubles(p.n, p.m, p.H, p.L) + (size_t)(wv * PPW + sub) * fnn_wave_scratch_doubles(p.n, p.m, p.H);
    fnn_stage_weights(p, smem);
    __syncthreads();
    const int stride = gridDim.x * FNN_W_WAVES * PPW;
    for (int i0 = (blockIdx.x * FNN_W_WAVES + wv) * PPW; i0 < p.batch; i0 += stride)   // (uniform trip count per wave: i0, not i0 + sub)
        fnn_jacobian_point<LW>(p, i0 + sub, hl, smem, y);
}

}  // namespace almpc
