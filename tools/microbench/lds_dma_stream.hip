// Microbenchmark (round 5, k_admm_inst<PACKED>): how fast do persistent 4-wave workgroups, two per CU, stream DISTINCT 58 KB chunks from
// HBM into LDS -- the packed triangle of one instance per chunk, workgroup b takes chunks b, b + grid, ...
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma_stream lds_dma_stream.hip && ./lds_dma_stream
// mode 0: global_load_lds_dwordx4 (no registers), wait, next chunk          mode 1: global_load_dwordx4 -> registers -> ds_write_b128
// mode 2: as 0 with TWO chunk buffers (the next chunk requested before the wait for the current one)
// mode 3: global_load_dwordx4 -> registers only (no LDS), the full-matrix kernel's way, same bytes
// Prints GB/s over the whole launch (chunks x bytes / time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int CHUNK = 58080;   // bytes: packed triangle of a 120 x 120 matrix

__global__ __launch_bounds__(256) void k(int mode, const char* src, int chunks, double* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double acc = 0.0;
    auto dma = [&](int c, char* dst) {
        const char* s = src + (size_t)c * CHUNK;
        const uint32_t base = (uint32_t)(uintptr_t)dst;
        for (int off = wv * 1024; off < CHUNK; off += 4 * 1024)
            if (off + lane * 16 < CHUNK)
                asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(s + off + lane * 16), "{m0}"(__builtin_amdgcn_readfirstlane(base + (uint32_t)off)) : "memory");
    };
    if (mode == 2 && (int)blockIdx.x < chunks) dma(blockIdx.x, smem);
    int par = 0;
    for (int c = blockIdx.x; c < chunks; c += gridDim.x) {
        if (mode == 0) {
            dma(c, smem);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            acc += reinterpret_cast<double*>(smem)[threadIdx.x];
            __syncthreads();
        } else if (mode == 2) {
            if (c + (int)gridDim.x < chunks) dma(c + gridDim.x, smem + (par ^ 1) * 58368);
            // wait for all but the newest request: 15 pieces per wave may stay in flight
            if (c + (int)gridDim.x < chunks) asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); else __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            acc += reinterpret_cast<double*>(smem + par * 58368)[threadIdx.x];
            __syncthreads();
            par ^= 1;
        } else {
            const char* s = src + (size_t)c * CHUNK;
            d2 v[15];
#pragma unroll
            for (int u = 0; u < 15; ++u) {
                const int off = (wv + 4 * u) * 1024 + lane * 16;
                v[u] = *reinterpret_cast<const d2*>(s + (off < CHUNK ? off : 0));
            }
            if (mode == 1) {
#pragma unroll
                for (int u = 0; u < 15; ++u) {
                    const int off = (wv + 4 * u) * 1024 + lane * 16;
                    if (off < CHUNK) *reinterpret_cast<d2*>(smem + off) = v[u];
                }
                __syncthreads();
                acc += reinterpret_cast<double*>(smem)[threadIdx.x];
                __syncthreads();
            } else {
#pragma unroll
                for (int u = 0; u < 15; ++u) acc += v[u][0] + v[u][1];
            }
        }
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

int main() {
    const int chunks = 4096 * 4;
    char* d; double* sink;
    hipMalloc(&d, (size_t)chunks * CHUNK); hipMalloc(&sink, 8);
    hipMemset(d, 0, (size_t)chunks * CHUNK);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 58368);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int per_cu : {2, 1}) for (int mode = 0; mode < 4; ++mode) {
        const size_t lds = (mode == 2) ? 2 * 58368 : 58368;
        if (mode == 2 && per_cu == 2) continue;
        const int grid = 256 * per_cu;
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, mode, d, chunks, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("workgroups per CU %d mode %d: %.3f ms, %.0f GB/s\n", per_cu, mode, best, (double)chunks * CHUNK / best / 1e6);
    }
    return 0;
}
