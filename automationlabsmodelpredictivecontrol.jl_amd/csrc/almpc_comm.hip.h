// almpc_comm.hip.h -- the multi-GPU side of the engine: RCCL over xGMI, inside the library (no torch, no MPI).
//
// Instances never interact (the reference has a single-instance API, src/main/computation_mpc.jl:17-55), so the batch is cut into
// contiguous shards, one process and one handle per GPU, and NO collective is on the data path of a step.
// What a multi-GPU caller still needs between steps is small and is what lives here:
//   almpc_comm_summary                 all-reduce of {ranks, unsolved instances} (sum) and {ADMM iterations, polish iterations} (max)
//   almpc_comm_allgather_first_input   all-gather of u[:,1] of every instance (m doubles each: 32 B per quadrotor instance, never the
//                                      full trajectories), what a plant simulator or logger on any rank needs for the next step
// librccl is opened with dlopen at the first use: libalmpc.so has no link-time dependency on it and single-GPU users never load it.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>

#include <string>

namespace almpc {

struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
    bool ok = false;
};

inline RcclApi& rccl_api() {
    static RcclApi api = []() {
        RcclApi a;
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* nm : names) {
            a.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (a.lib) break;
        }
        if (!a.lib) { a.err = std::string("dlopen(librccl.so): ") + (dlerror() ? dlerror() : "not found"); return a; }
        auto sym = [&](const char* s) { void* p = dlsym(a.lib, s); if (!p && a.err.empty()) a.err = std::string("librccl.so lacks ") + s; return p; };
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(sym("ncclAllGather"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
        a.ok = a.err.empty();
        return a;
    }();
    return api;
}

// out[0] = 1 (this rank), out[1] = instances with status != 0, out[2] = max ADMM iterations, out[3] = max polish iterations
inline __global__ __launch_bounds__(1024) void k_comm_summary(int batch, const int32_t* status, const int32_t* iters, const int32_t* piters,
                                                       long long* out) {
    __shared__ long long sh[3][16];
    long long bad = 0, mi = 0, mp = 0;
    for (int i = threadIdx.x; i < batch; i += blockDim.x) {
        bad += status[i] != 0 ? 1 : 0;
        mi = iters[i] > mi ? iters[i] : mi;
        mp = piters[i] > mp ? piters[i] : mp;
    }
    for (int o = 32; o > 0; o >>= 1) {
        bad += __shfl_xor(bad, o);
        const long long a = __shfl_xor(mi, o), b = __shfl_xor(mp, o);
        mi = a > mi ? a : mi;
        mp = b > mp ? b : mp;
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][wv] = bad; sh[1][wv] = mi; sh[2][wv] = mp; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bad = 0; mi = 0; mp = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
            bad += sh[0][w];
            mi = sh[1][w] > mi ? sh[1][w] : mi;
            mp = sh[2][w] > mp ? sh[2][w] : mp;
        }
        out[0] = 1; out[1] = bad; out[2] = mi; out[3] = mp;
    }
}

// dst[i][a] = u[i][0][a]: the first input of every instance, packed for the all-gather
inline __global__ __launch_bounds__(256) void k_pack_first_input(int batch, int m, int N, const double* u, double* dst) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < (long)batch * m) dst[t] = u[(t / m) * (long)m * N + (t % m)];
}

}  // namespace almpc
