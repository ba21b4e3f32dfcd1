// Microbenchmark of k_design_inverse_wave<NCOL> (one wave per matrix): B SPD matrices of size nz, time per launch and the residual
// max |A Ainv - I|.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I automationlabsmodelpredictivecontrol.jl_amd/csrc \
//                         -o /tmp/inverse_wave tools/microbench/inverse_wave.hip && /tmp/inverse_wave [nz] [batch]
#include "almpc_kernels.hip.h"
#include "almpc_design.hip.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
using namespace almpc;
int main(int argc, char** argv) {
    const int nz = argc > 1 ? atoi(argv[1]) : 40, b = argc > 2 ? atoi(argv[2]) : 1024;
    const int nzs = ((nz + 15) / 16) * 16;
    std::vector<double> A((size_t)b * nz * nzs, 0.0);
    srand(1);
    for (int q = 0; q < b; ++q) {
        std::vector<double> M((size_t)nz * nz);
        for (auto& v : M) v = rand() / (double)RAND_MAX - 0.5;
        std::vector<double> H((size_t)nz * nz, 0.0);
        for (int i = 0; i < nz; ++i) for (int j = 0; j < nz; ++j) { double s = 0; for (int k = 0; k < nz; ++k) s += M[i * nz + k] * M[j * nz + k]; H[i * nz + j] = s + (i == j ? 0.5 : 0.0); }
        for (int i = 0; i < nz; ++i) for (int j = 0; j < nz; ++j) A[(size_t)q * nz * nzs + (size_t)j * nzs + i] = H[i * nz + j] / sqrt(H[i * nz + i] * H[j * nz + j]);
    }
    double *dA, *dO; int* dF;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dO, A.size() * 8); hipMalloc(&dF, b * 4);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemset(dO, 0, A.size() * 8); hipMemset(dF, 0, b * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        for (int it = 0; it < 20; ++it)
            launch_design_inverse(dim3(1, b), 520 * 8, 0, nz, nzs, dA, 0.0, nullptr, dO, dF, (long)nz * nzs, 0L, (long)nz * nzs, 1L);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("nz %d batch %d: %.1f us per launch\n", nz, b, 1e3 * ms / 20);
    }
#ifdef INV_STAMPS
    { long long d[16]; hipMemcpyFromSymbol(d, HIP_SYMBOL(g_inv_dbg), sizeof(d));
      printf("cycles: loads %lld, pivots %lld (%.0f per pivot), stores %lld\n", d[1] - d[0], d[2] - d[1], (double)(d[2] - d[1]) / nz, d[3] - d[2]);
      printf("  blocks of 8 pivots:"); for (int q = 0; q * 8 < nz; ++q) printf(" %lld", (q * 8 + 8 < nz ? d[5 + q] : d[2]) - d[4 + q]); printf("\n");
      if (nz > 64) printf("  four-wave kernel, pivot 5, wave 0: barrier %lld, reads + f %lld, next column + publication %lld, other columns %lld\n", d[10] - d[9], d[11] - d[10], d[12] - d[11], d[13] - d[12]);
      else printf("  pivot 11: w[k+1] + fma + publish %lld, readlane %lld, reciprocal %lld, other columns %lld\n", d[11] - d[10], d[12] - d[11], d[13] - d[12], d[14] - d[13]); }
#endif
    std::vector<double> O(A.size()); std::vector<int> F(b);
    hipMemcpy(O.data(), dO, O.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(F.data(), dF, b * 4, hipMemcpyDeviceToHost);
    double worst = 0; int nf = 0;
    for (int q = 0; q < b; q += 37) {
        const double* a = &A[(size_t)q * nz * nzs]; const double* o = &O[(size_t)q * nz * nzs];
        for (int i = 0; i < nz; ++i) for (int j = 0; j < nz; ++j) { double s = 0; for (int k = 0; k < nz; ++k) s += a[(size_t)k * nzs + i] * o[(size_t)j * nzs + k]; worst = fmax(worst, fabs(s - (i == j))); }
    }
    for (int q = 0; q < b; ++q) nf += F[q] != 0;
    printf("max |A Ainv - I| %.2e, flagged %d\n", worst, nf);
    return 0;
}
