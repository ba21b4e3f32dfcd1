// One stage of the k_sdual recursion in isolation (one wave, s_memtime based): y = M x for a 16 x 16 matrix held one row per lane,
// x distributed one element per lane of a row of 16, chained 256 times (x <- y).  Variants of the broadcast of x (one kernel each):
//   0  v_fmac_f64_dpp row_newbcast (no LDS, no SGPR)          1  LDS: ds_write_b64, fence, 8 uniform ds_read_b128, v_fma_f64
//   2  v_readlane pairs -> SGPR operands of v_fma_f64          3  chain of 16 plain v_fma_f64 (no broadcast: the FMA floor)
//   4  as 0 plus the LDS traffic of the kernel's forward stage (one ds_read_b64 for the next stage, one ds_write_b64 of the result)
//   5  as 4 with the read consumed before the write is issued
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dpp_matvec tools/microbench/dpp_matvec.hip && /tmp/dpp_matvec
#include <hip/hip_runtime.h>
#include <cstdio>
template <int J, bool FIRST>
__device__ __forceinline__ void fma_bcast(double& acc, double src, double coef) {
    if constexpr (FIRST) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(coef), "n"(J));
    else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(coef), "n"(J));
}
template <int IDX = 0>
__device__ __forceinline__ void dot16(double& a0, double& a1, double src, const double* c) {
    if constexpr (IDX < 16) {
        if constexpr ((IDX & 1) == 0) fma_bcast<IDX, IDX == 0>(a0, src, c[IDX]); else fma_bcast<IDX, false>(a1, src, c[IDX]);
        dot16<IDX + 1>(a0, a1, src, c);
    }
}
__device__ __forceinline__ double readlane_d(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ void fence_w() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
template <int MODE>
__global__ void k(const double* M, double* out, long long* cyc) {
    __shared__ __attribute__((aligned(16))) double buf[64 * 260];
    const int lane = threadIdx.x & 63, l16 = lane & 15;
    double c[16];
    for (int j = 0; j < 16; ++j) c[j] = M[l16 * 16 + j];
    double x = 1.0 + 0.01 * l16;
    for (int t = lane; t < 64 * 260; t += 64) buf[t] = 1e-3 * (t & 7);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 256; ++it) {
        double a0 = 0.0, a1 = 0.0;
        if constexpr (MODE == 0) dot16(a0, a1, x, c);
        else if constexpr (MODE == 1) {
            if (lane < 16) buf[lane] = x;
            fence_w();
            double g[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) g[j] = buf[j];
#pragma unroll
            for (int j = 0; j < 16; j += 2) { a0 = __builtin_fma(c[j], g[j], a0); a1 = __builtin_fma(c[j + 1], g[j + 1], a1); }
            fence_w();
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 16; j += 2) { a0 = __builtin_fma(c[j], readlane_d(x, j), a0); a1 = __builtin_fma(c[j + 1], readlane_d(x, j + 1), a1); }
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 16; j += 2) { a0 = __builtin_fma(c[j], x, a0); a1 = __builtin_fma(c[j + 1], x, a1); }
        } else if constexpr (MODE == 4) {
            const double nxt = buf[(it + 1) * 16 + l16];
            dot16(a0, a1, x, c);
            if (lane < 16) buf[(it + 2) * 16 + l16 + 64 * 130] = a0 + a1;
            a0 += (l16 >= 12) ? nxt : 0.0;
        } else {   // the same with the read consumed before the write is issued
            const double nxt = buf[(it + 1) * 16 + l16];
            dot16(a0, a1, x, c);
            const double o = a0 + a1;
            x = (l16 >= 12) ? nxt : o;
            asm volatile("" :: "v"(x) : "memory");
            if (lane < 16) buf[(it + 2) * 16 + l16 + 64 * 130] = o;
            continue;
        }
        x = a0 + a1;
    }
    const long long t1 = __builtin_readcyclecounter();
    out[MODE * 64 + lane] = x;
    if (lane == 0) cyc[MODE] = t1 - t0;
}
int main() {
    double h[256];
    for (int i = 0; i < 256; ++i) h[i] = ((i * 37) % 17 - 8) / 70.0;
    double *dM, *dout; long long* dc;
    (void)hipMalloc(&dM, sizeof(h)); (void)hipMalloc(&dout, 6 * 64 * 8); (void)hipMalloc(&dc, 6 * 8);
    (void)hipMemcpy(dM, h, sizeof(h), hipMemcpyHostToDevice);
    const char* names[6] = {"v_fmac_f64_dpp row_newbcast", "LDS write + uniform reads + v_fma_f64", "v_readlane -> SGPR operand", "plain v_fma_f64 (floor)",
                            "dpp + one ds_read + one ds_write per stage", "  ... read consumed before the write"};
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dM, dout, dc); hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dM, dout, dc);
        hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, dM, dout, dc); hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, dM, dout, dc);
        hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, dM, dout, dc); hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, dM, dout, dc);
    }
    (void)hipDeviceSynchronize();
    long long c[6]; double o[384];
    (void)hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost); (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    for (int mode = 0; mode < 6; ++mode) printf("%-45s %6.1f cycles per 16 x 16 stage   (x[3] = %.12g)\n", names[mode], c[mode] / 256.0, o[mode * 64 + 3]);
    return 0;
}
