/* Sanitizer driver for the CPU restatement (oracle/almpc_oracle.c): compiled by tests/test_sanitizers.py with
 * -fsanitize=address,undefined -fno-sanitize-recover=all; reads the arrays the test wrote (raw little-endian doubles), runs
 * almpc_oracle_step_batch single-threaded and writes u back.  Test infrastructure only. */
#include "../../oracle/almpc_oracle.c"

#include <stdio.h>

static double* rd(FILE* f, size_t cnt) {
    double* v = (double*)malloc((cnt ? cnt : 1) * sizeof(double));
    if (fread(v, sizeof(double), cnt, f) != cnt) { fprintf(stderr, "short input\n"); exit(2); }
    return v;
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    double hdr[8];
    if (fread(hdr, sizeof(double), 8, f) != 8) return 2;
    const int n = (int)hdr[0], m = (int)hdr[1], N = (int)hdr[2], batch = (int)hdr[3], max_iter = (int)hdr[4];
    const double sigma = hdr[5];
    const int nz = m * N;
    double* A = rd(f, (size_t)n * n); double* B = rd(f, (size_t)n * m); double* Minv = rd(f, (size_t)nz * nz);
    double* Hs = rd(f, (size_t)nz * nz); double* G = rd(f, (size_t)nz * nz); double* Fs = rd(f, (size_t)nz * n);
    double* fS = rd(f, nz); double* lo = rd(f, nz); double* hi = rd(f, nz); double* d = rd(f, nz);
    double* xref = rd(f, (size_t)n * (N + 1)); double* uref = rd(f, (size_t)nz); double* x0 = rd(f, (size_t)batch * n);
    double* rho = rd(f, nz);
    fclose(f);
    double* x = (double*)malloc(sizeof(double) * batch * n * (N + 1)); double* ex = (double*)malloc(sizeof(double) * batch * n * (N + 1));
    double* u = (double*)malloc(sizeof(double) * batch * nz); double* eu = (double*)malloc(sizeof(double) * batch * nz);
    int32_t* st = (int32_t*)malloc(sizeof(int32_t) * batch); int32_t* it = (int32_t*)malloc(sizeof(int32_t) * batch);
    int32_t* pit = (int32_t*)malloc(sizeof(int32_t) * batch);
    const int used = almpc_oracle_step_batch(n, m, N, batch, A, B, Minv, Hs, G, Fs, fS, lo, hi, d, xref, uref, x0, rho, sigma, 1.6, 1e-3, 1e-3,
                                             max_iter, max_iter, 1, 0, x, ex, u, eu, st, it, pit, 1);
    if (used < 0) return 3;
    FILE* o = fopen(argv[2], "wb");
    fwrite(u, sizeof(double), (size_t)batch * nz, o);
    fclose(o);
    int bad = 0;
    for (int b = 0; b < batch; ++b) bad += st[b] != 0;
    printf("instances %d unsolved %d\n", batch, bad);
    free(A); free(B); free(Minv); free(Hs); free(G); free(Fs); free(fS); free(lo); free(hi); free(d); free(xref); free(uref); free(x0); free(rho);
    free(x); free(ex); free(u); free(eu); free(st); free(it); free(pit);
    return 0;
}
