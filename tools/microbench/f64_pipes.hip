// Microbenchmark: FP64 MFMA rate, FP64 VALU FMA rate, and whether the two pipes overlap on one SIMD (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o f64_pipes f64_pipes.hip && ./f64_pipes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

// mode 0: every wave MFMA; mode 1: every wave VALU FMA; mode 2: even waves MFMA, odd waves VALU
__global__ __launch_bounds__(512) void k(int mode, int iters, double* out, long long* cyc) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = (mode == 0) || (mode == 2 && (wave & 1) == 0) || (mode == 3 && (wave >> 2) == 0);
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    long long t0 = __builtin_readcyclecounter();
    double res = 0;
    if (do_mfma) {
        d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        res = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        double x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // 32 FMAs per iteration
                x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
                x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
            }
        }
        res = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

int main() {
    const int blocks = 256, iters = 20000;
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * 512);
    hipMalloc(&cyc, sizeof(long long) * blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512}) {
        for (int mode = 0; mode < 4; ++mode) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, mode, 100, out, cyc);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, mode, iters, out, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const int waves = threads / 64;
            double mf = 0, vf = 0;
            for (int w = 0; w < waves; ++w) {
                bool m = (mode == 0) || (mode == 2 && (w & 1) == 0) || (mode == 3 && (w >> 2) == 0);
                if (m) mf += 4.0 * 2048 * iters; else vf += 32.0 * 128 * iters;
            }
            std::vector<long long> hc(blocks * waves);
            hipMemcpy(hc.data(), cyc, sizeof(long long) * blocks * waves, hipMemcpyDeviceToHost);
            printf("threads %d mode %d: %.3f ms  MFMA %.1f TF  VALU %.1f TF  total %.1f TF  | wave0 cycles(100MHz ticks?) %lld per-iter %.1f\n",
                   threads, mode, ms, mf * blocks / ms / 1e9, vf * blocks / ms / 1e9, (mf + vf) * blocks / ms / 1e9,
                   hc[0], (double)hc[0] / iters); printf("   last wave per-iter %.1f\n", (double)hc[waves-1]/iters);
        }
    }
    return 0;
}
