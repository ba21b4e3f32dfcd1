// Microbenchmark: how long 256 workgroups (one per CU, 8 waves) take to pull the SAME shared operands out of L2 at kernel start --
// the prologue of k_step_fused: 115 KB of MFMA A-fragments into registers + 115 KB of G into LDS, every workgroup the same bytes.
//   hipcc --offload-arch=gfx950 -O3 -o prologue_loads prologue_loads.hip && ./prologue_loads
// mode 0: fragments, 30 x 8 B per lane (global_load_dwordx2)          mode 1: fragments, 15 x 16 B per lane (dwordx4)
// mode 2: G rows straight into LDS (global_load_lds_dwordx4)          mode 3: 0 + 2 (the fused prologue)
// mode 4: as 0, wave w of workgroup b takes row block (w + b/8) % 8   mode 5: as 2, rows start at (b/8) * 15 and wrap
// mode 6: 4 + 5                                                       mode 7: 1 + 5
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int NZ = 120, KS = 30, NRB = 8;

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k(int mode, const double* frag, const double* G, double* out, long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long t0 = __builtin_readcyclecounter();
    double acc = 0.0;
    const int rot = (int)(blockIdx.x >> 3);
    const bool fr8 = mode == 0 || mode == 3 || mode == 4 || mode == 6;
    const bool fr16 = mode == 1 || mode == 7;
    const bool lds = mode == 2 || mode == 3 || mode == 5 || mode == 6 || mode == 7;
    const bool rotf = mode == 4 || mode == 6;
    const bool rotg = mode == 5 || mode == 6 || mode == 7;
    double a[KS];
    d2 a2[KS / 2];
    if (fr8) {
        const int rb = rotf ? ((wv + rot) & 7) : wv;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] = frag[((size_t)(rb * KS + ks)) * 64 + lane];
    }
    if (fr16) {
#pragma unroll
        for (int j = 0; j < KS / 2; ++j) a2[j] = *reinterpret_cast<const d2*>(frag + ((size_t)(wv * (KS / 2) + j)) * 128 + 2 * lane);
    }
    if (lds) {
        const int r0 = rotg ? (rot * 15) % NZ : 0;
        for (int i = wv; i < NZ; i += NRB) {
            int r = i + r0; if (r >= NZ) r -= NZ;
            if (lane < NZ / 2) {
                const char* src = reinterpret_cast<const char*>(G + (size_t)r * 128) + lane * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(smem + (size_t)r * NZ), 16, 0, 0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (fr8) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc += a[ks];
    }
    if (fr16) {
#pragma unroll
        for (int j = 0; j < KS / 2; ++j) acc += a2[j][0] + a2[j][1];
    }
    const long long t2 = __builtin_readcyclecounter();   // fragments have arrived
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const long long t3 = __builtin_readcyclecounter();   // everything has arrived, for every wave
    if (lds) acc += smem[(threadIdx.x * 7) % (NZ * NZ)];
    out[blockIdx.x * 512 + threadIdx.x] = acc;
    if (lane == 0) {
        long long* c = cyc + ((size_t)blockIdx.x * 8 + wv) * 4;
        c[0] = t1 - t0; c[1] = t2 - t0; c[2] = t3 - t0;
    }
}

int main() {
    const int blocks = 256;
    double *frag, *G, *out; long long* cyc;
    hipMalloc(&frag, sizeof(double) * NRB * KS * 64 + 4096);
    hipMalloc(&G, sizeof(double) * NZ * 128);
    hipMalloc(&out, sizeof(double) * blocks * 512);
    hipMalloc(&cyc, sizeof(long long) * blocks * 8 * 4);
    hipMemset(frag, 0, sizeof(double) * NRB * KS * 64 + 4096);
    hipMemset(G, 0, sizeof(double) * NZ * 128);
    const size_t lds = (size_t)NZ * NZ * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int mode = 0; mode < 8; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), lds, 0, mode, frag, G, out, cyc);
            hipDeviceSynchronize();
        }
        std::vector<long long> hc(blocks * 8 * 4);
        hipMemcpy(hc.data(), cyc, sizeof(long long) * hc.size(), hipMemcpyDeviceToHost);
        for (int q = 0; q < 3; ++q) {
            std::vector<long long> v;
            for (int i = 0; i < blocks * 8; ++i) v.push_back(hc[(size_t)i * 4 + q]);
            std::sort(v.begin(), v.end());
            printf("mode %d  %s: median %6lld  p90 %6lld  max %6lld\n", mode, q == 0 ? "issued   " : (q == 1 ? "fragments" : "all+barr."),
                   v[v.size() / 2], v[v.size() * 9 / 10], v.back());
        }
    }
    return 0;
}
