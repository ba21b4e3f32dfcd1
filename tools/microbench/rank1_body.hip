// The inner body of the register Gauss-Jordan kernels in isolation: NR uniform ds_read_b128 (2 NR pivot-row entries) + 2 NR x ROWS FMAs with
// three VGPR sources, one wave, no dependence between iterations except through the accumulators.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/rank1_body tools/microbench/rank1_body.hip && /tmp/rank1_body
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int NR, int ROWS, int MODE>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc) {
    __shared__ __attribute__((aligned(16))) double w[4][2 * NR + 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = lane; i < 2 * NR; i += 64) w[wv][i] = 1e-3 * i;
    double S[ROWS][2 * NR];
#pragma unroll
    for (int a = 0; a < ROWS; ++a)
#pragma unroll
        for (int c = 0; c < 2 * NR; ++c) S[a][c] = lane + c + a;
    double f[ROWS];
#pragma unroll
    for (int a = 0; a < ROWS; ++a) f[a] = 1e-6 * (lane + a);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    d2 wreg[NR];
#pragma unroll
    for (int c = 0; c < NR; ++c) wreg[c] = *reinterpret_cast<const d2*>(&w[wv][2 * c]);
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
        if (MODE == 1) __syncthreads();
#pragma unroll
        for (int c = 0; c < 2 * NR; c += 2) {
            const d2 wc = MODE == 2 ? wreg[c / 2] : *reinterpret_cast<const d2*>(&w[wv][c]);
#pragma unroll
            for (int a = 0; a < ROWS; ++a) {
                S[a][c] = __builtin_fma(-f[a], wc[0], S[a][c]);
                S[a][c + 1] = __builtin_fma(-f[a], wc[1], S[a][c + 1]);
            }
        }
        if (MODE == 2) {   // FMAs only: w stays in registers
        }
#pragma unroll
        for (int a = 0; a < ROWS; ++a) f[a] += 1e-9;
        if (MODE == 2) {
#pragma unroll
            for (int c = 0; c < NR; ++c) wreg[c][0] += 1e-9;
        } else {
            if (lane == 0) w[wv][it & 7] = f[0];   // keep the reads in the loop
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double r = 0;
#pragma unroll
    for (int a = 0; a < ROWS; ++a)
#pragma unroll
        for (int c = 0; c < 2 * NR; ++c) r += S[a][c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NR, int ROWS, int MODE>
void run(const char* name, int threads, double* out, long long* cyc) {
    long long h = 0;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<NR, ROWS, MODE>), dim3(1), dim3(threads), 0, 0, out, cyc); hipDeviceSynchronize(); }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-60s %d waves: %.0f cycles per iteration (%d reads, %d FMAs)\n", name, threads / 64, h / 256.0, NR, 2 * NR * ROWS);
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 8);
    run<24, 1, 0>("one row x 48 columns (one-wave kernel)", 64, out, cyc);
    run<24, 1, 0>("one row x 48 columns (one-wave kernel)", 256, out, cyc);
    run<16, 2, 0>("two rows x 32 columns (four-wave kernel)", 64, out, cyc);
    run<16, 2, 0>("two rows x 32 columns (four-wave kernel)", 256, out, cyc);
    run<16, 2, 1>("two rows x 32 columns + __syncthreads", 256, out, cyc);
    run<16, 2, 2>("two rows x 32 columns, pivot row kept in registers", 64, out, cyc);
    run<16, 2, 2>("two rows x 32 columns, pivot row kept in registers", 256, out, cyc);
    run<8, 4, 0>("four rows x 16 columns", 256, out, cyc);
    run<4, 8, 0>("eight rows x 8 columns", 256, out, cyc);
    return 0;
}
