// Dependent-chain latencies of one wave on gfx950 (cycles per link of a 512-link chain, s_memtime based):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dep_latency tools/microbench/dep_latency.hip && /tmp/dep_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double readlane_d(double v, int l) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__global__ void k(int mode, double seed, double* out, long long* cyc, double* relerr) {
    __shared__ double buf[128];
    const int lane = threadIdx.x & 63;
    double x = seed + lane * 1e-3, y = 0.999 + lane * 1e-6;
    buf[lane] = x; buf[64 + lane] = y;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (mode == 0) {            // v_fma_f64 -> v_fma_f64
#pragma unroll 16
        for (int i = 0; i < 512; ++i) x = __builtin_fma(x, y, 1e-9);
    } else if (mode == 1) {     // v_rcp_f64 -> v_rcp_f64
#pragma unroll 16
        for (int i = 0; i < 512; ++i) x = __builtin_amdgcn_rcp(x) + 0.0 * y;
    } else if (mode == 2) {     // v_readlane pair -> v_fma_f64 (SGPR operand)
#pragma unroll 16
        for (int i = 0; i < 512; ++i) x = __builtin_fma(readlane_d(x, i & 63), y, 1e-9);
    } else if (mode == 3) {     // ds_write_b64 -> ds_read_b64 of another lane's value
#pragma unroll 16
        for (int i = 0; i < 512; ++i) {
            buf[lane] = x;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            x = buf[(lane + 1) & 63] * y;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    } else if (mode == 4) {     // v_mul_f32 -> v_mul_f32
        float a = (float)x, b = (float)y;
#pragma unroll 16
        for (int i = 0; i < 512; ++i) a = __builtin_fmaf(a, b, 1e-9f);
        x = a;
    } else if (mode == 5) {     // 8 independent FMA chains (throughput)
        double z[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) z[j] = x + j;
#pragma unroll 4
        for (int i = 0; i < 512; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = __builtin_fma(z[j], y, 1e-9);
        x = z[0] + z[1] + z[2] + z[3] + z[4] + z[5] + z[6] + z[7];
    } else if (mode == 6) {     // reciprocal + two Newton steps, dependent
#pragma unroll 8
        for (int i = 0; i < 512; ++i) {
            double r = __builtin_amdgcn_rcp(x);
            r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
            r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
            x = r + 0.5;
        }
    } else if (mode == 7) {     // ds_read_b128 at a wave-uniform address, 8 independent reads per link (return-path throughput)
        double acc = 0.0;
#pragma unroll 4
        for (int i = 0; i < 512; ++i) {
            const d2* p = reinterpret_cast<const d2*>(buf + ((i & 3) << 4));
#pragma unroll
            for (int j = 0; j < 8; ++j) { const d2 v = p[j]; acc += v[0] * y + v[1]; }
        }
        x = acc;
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (mode == 8) {            // accuracy of v_rcp_f64
        double worst = 0.0;
        for (int i = 0; i < 4096; ++i) {
            const double v = seed * (1.0 + (lane * 4096 + i) * 3.7e-6);
            const double r = __builtin_amdgcn_rcp(v);
            const double e = fabs(__builtin_fma(-v, r, 1.0));
            worst = e > worst ? e : worst;
        }
        relerr[threadIdx.x] = worst;
    }
}
int main() {
    double* out; long long* cyc; double* re;
    hipMalloc(&out, 8 * 4096); hipMalloc(&cyc, 8 * 64); hipMalloc(&re, 8 * 256);
    const char* names[] = {"v_fma_f64 chain", "v_rcp_f64 (+add) chain", "readlane pair -> fma chain", "ds_write -> fence -> ds_read chain (x2 fences)", "v_fma_f32 chain",
                           "8 independent fma_f64 per link", "rcp + 2 Newton steps (+add) chain", "8 x ds_read_b128 uniform address + 16 FMA per link"};
    for (int mode = 0; mode < 8; ++mode)
        for (int waves = 1; waves <= 8; waves *= 8) {
            long long h = 0;
            for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 0, 0, mode, 1.25, out, cyc, re); hipDeviceSynchronize(); }
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("%-52s %d wave(s)/CU: %.1f cycles per link\n", names[mode], waves, h / 512.0);
        }
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 8, 1.25, out, cyc, re); hipDeviceSynchronize();
    double hr[64]; hipMemcpy(hr, re, 8 * 64, hipMemcpyDeviceToHost);
    double w = 0; for (int i = 0; i < 64; ++i) w = fmax(w, hr[i]);
    printf("v_rcp_f64: max |1 - v r| = %.3e = 2^%.1f\n", w, log2(w));
    return 0;
}
