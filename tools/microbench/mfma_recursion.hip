// Microbenchmark: latency of the state recursion e+ = [A|B][e; v] on the FP64 matrix cores, one wave, 16 columns.
//   hipcc --offload-arch=gfx950 -O3 mfma_recursion.hip -o mfma_recursion && ./mfma_recursion
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>  // 0: products only; 1: + LDS operand per stage; 2: + 8 global stores per stage; 3: stores without products
__global__ void k(double* out, long long* cyc, int N, const double* A) {
    __shared__ double Ut[30 * 4 * 17];
    const int lane = threadIdx.x, q = lane >> 4, col = lane & 15;
    for (int t = lane; t < 30 * 4 * 17; t += 64) Ut[t] = 1e-3 * t;
    double afr[3], bfr;
    for (int s = 0; s < 3; ++s) afr[s] = A[(4 * s + q) * 16 + col];
    bfr = A[(12 + q) * 16 + col];
    d4 acc = {1.0, 2.0, 3.0, 0.0};
    __syncthreads();
    double* o = out + (size_t)(blockIdx.x * 16 + col) * 372 + 12;
    const long long t0 = __builtin_readcyclecounter();
    for (int kk = 0; kk < N; ++kk) {
        d4 nx = {0.0, 0.0, 0.0, 0.0};
        const double u = (MODE >= 1) ? Ut[(kk * 4 + q) * 17 + col] : 0.5;
        if (MODE != 3) {
            nx = __builtin_amdgcn_mfma_f64_16x16x4f64(bfr, u, nx, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 3; ++s) nx = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[s], acc[s], nx, 0, 0, 0);
        } else {
            nx = acc * 0.999 + u;
        }
        acc = nx;
        if (MODE >= 2) {
#pragma unroll
            for (int reg = 0; reg < 3; ++reg) {
                o[kk * 12 + q + 4 * reg] = acc[reg];
                o[kk * 12 + q + 4 * reg + 4096 * 372] = acc[reg] + 1.0;
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc[0] == 1.2345) out[0] = acc[1];
}
int main() {
    double *out, *A; long long* cyc;
    hipMalloc(&out, sizeof(double) * 2 * 4096 * 372 + 1024); hipMalloc(&A, 256 * 8); hipMalloc(&cyc, 256 * 8);
    std::vector<double> hA(256, 0.01); hipMemcpy(A, hA.data(), 256 * 8, hipMemcpyHostToDevice);
    std::vector<long long> h(256);
    for (int grid : {1, 256}) {
        for (int mode = 0; mode < 4; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, out, cyc, 30, A);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, out, cyc, 30, A);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, out, cyc, 30, A);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(64), 0, 0, out, cyc, 30, A);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
            long long mx = 0, sum = 0; for (int i = 0; i < grid; ++i) { mx = h[i] > mx ? h[i] : mx; sum += h[i]; }
            printf("grid %3d mode %d: 30 stages, cycles (s_memtime units) mean %lld max %lld -> %.0f per stage\n", grid, mode, sum / grid, mx, (double)sum / grid / 30);
        }
    }
    return 0;
}
