/*
 * capi_harness.c -- a caller of include/almpc.h that is neither Python nor ctypes: plain C, linked against libalmpc.so, passing
 * column-major arrays exactly as Julia's `ccall` passes a Matrix{Float64} (julia/AlmpcHIP.jl), i.e. the way the reference would
 * reach the library through proceed_controller / update_initialization! / calculate! (src/main/main_mpc.jl:22-53,
 * src/main/computation_mpc.jl:17-55).  Test infrastructure: tests/test_gpu_capi_harness.py compiles it with gcc and runs it on
 * the golden vectors of tests/golden/ (it reads a small text file the test writes: the problem, the initial states and the
 * expected inputs / states).
 *
 *   usage: capi_harness <case-file>          exit 0 and a line "OK ..." if max |u - u_golden| <= 1e-6 and max |x - x_golden| <= 1e-5
 *
 * file format (whitespace separated):  n m N batch  A[n*n col-major] B[n*m] Q[n*n] R[m*m] umin[m] umax[m]
 *                                      xref[n*(N+1) col-major] uref[m*N]  x0[batch][n]  u[batch][N][m]  x[batch][N+1][n]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/almpc.h"

static double* read_vec(FILE* f, size_t cnt) {
    double* v = (double*)malloc((cnt ? cnt : 1) * sizeof(double));
    for (size_t i = 0; i < cnt; ++i)
        if (fscanf(f, "%lf", &v[i]) != 1) { fprintf(stderr, "short case file\n"); exit(2); }
    return v;
}

#define CHECK(call)                                                                           \
    do {                                                                                      \
        int rc_ = (call);                                                                     \
        if (rc_ != ALMPC_OK) {                                                                \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, h ? almpc_last_error(h) : "");      \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <case-file>\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "r");
    if (!f) { perror(argv[1]); return 2; }
    int n, m, N, batch;
    if (fscanf(f, "%d %d %d %d", &n, &m, &N, &batch) != 4) return 2;
    double* A = read_vec(f, (size_t)n * n);
    double* B = read_vec(f, (size_t)n * m);
    double* Q = read_vec(f, (size_t)n * n);
    double* R = read_vec(f, (size_t)m * m);
    double* umin = read_vec(f, m);
    double* umax = read_vec(f, m);
    double* xref = read_vec(f, (size_t)n * (N + 1));
    double* uref = read_vec(f, (size_t)m * N);
    double* x0 = read_vec(f, (size_t)batch * n);
    double* ug = read_vec(f, (size_t)batch * N * m);
    double* xg = read_vec(f, (size_t)batch * (N + 1) * n);
    fclose(f);

    almpc_handle* h = NULL;
    CHECK(almpc_create(&h, n, m, N, batch, 0, 0));
    /* S = NULL (0), P = NULL (DARE inside, src/sub/design_mpc.jl:327), no state box, OSQP's rho / sigma */
    CHECK(almpc_design_shared(h, A, B, Q, R, NULL, NULL, umin, umax, NULL, NULL, 0.1, 1e-6));
    CHECK(almpc_set_reference(h, xref, uref, 0));
    CHECK(almpc_update_initialization(h, x0));            /* update_initialization!(C, x0) */
    almpc_opts o;
    almpc_default_opts(&o);
    CHECK(almpc_calculate(h, &o));                        /* calculate!(C) */
    double* u = (double*)malloc((size_t)batch * N * m * sizeof(double));
    double* eu = (double*)malloc((size_t)batch * N * m * sizeof(double));
    double* x = (double*)malloc((size_t)batch * (N + 1) * n * sizeof(double));
    double* ex = (double*)malloc((size_t)batch * (N + 1) * n * sizeof(double));
    int32_t* status = (int32_t*)malloc((size_t)batch * sizeof(int32_t));
    CHECK(almpc_get_results(h, x, ex, u, eu, status, NULL, NULL));
    double eu_max = 0.0, ex_max = 0.0, dev_max = 0.0;
    int bad = 0;
    for (int i = 0; i < batch; ++i) bad += status[i] != ALMPC_SOLVED;
    for (size_t t = 0; t < (size_t)batch * N * m; ++t) {
        eu_max = fmax(eu_max, fabs(u[t] - ug[t]));
        dev_max = fmax(dev_max, fabs(eu[t] - (u[t] - uref[t % ((size_t)N * m)])));   /* e_u = u - u_reference (..linear.jl:84-87) */
    }
    for (size_t t = 0; t < (size_t)batch * (N + 1) * n; ++t) {
        ex_max = fmax(ex_max, fabs(x[t] - xg[t]));
        dev_max = fmax(dev_max, fabs(ex[t] - (x[t] - xref[t % ((size_t)(N + 1) * n)])));
    }
    /* the handle's terminal weight is what the reference would store in TerminalIngredient.P */
    double* P = (double*)malloc((size_t)n * n * sizeof(double));
    CHECK(almpc_get_design(h, NULL, NULL, P, NULL));
    /* multi-GPU entry points with a one-rank communicator: the same calls a Julia process per GPU would make */
    char id[ALMPC_COMM_ID_BYTES];
    int64_t summ[4] = {0, 0, 0, 0};
    int comm_rc = almpc_comm_unique_id(id);
    if (comm_rc == ALMPC_OK) {
        CHECK(almpc_comm_init(h, id, 0, 1));
        CHECK(almpc_comm_summary(h, summ));
        double* u0 = (double*)malloc((size_t)batch * m * sizeof(double));
        CHECK(almpc_comm_allgather_first_input(h, u0, NULL));
        for (int i = 0; i < batch; ++i)
            for (int a = 0; a < m; ++a)
                if (u0[(size_t)i * m + a] != u[(size_t)i * N * m + a]) { fprintf(stderr, "gathered first input differs\n"); return 1; }
        free(u0);
    }
    almpc_destroy(h);
    /* The same problems as ONE MODEL PER INSTANCE with a state box (kw mpc_state_constraint of the black-box delegates,
     * .../fnn/mpc_modeler_implementation_fnn.jl:52-58) wide enough to stay inactive: almpc_set_state_box + almpc_design_batched run the
     * per-instance constraint-space build and the state-row finish, and must land on the same golden inputs. */
    double eu2_max = 0.0;
    int bad2 = 0;
    {
        h = NULL;
        CHECK(almpc_create(&h, n, m, N, batch, 0, 0));
        double* Ab = (double*)malloc((size_t)batch * n * n * sizeof(double));
        double* Bb = (double*)malloc((size_t)batch * n * m * sizeof(double));
        for (int i = 0; i < batch; ++i) {
            memcpy(Ab + (size_t)i * n * n, A, (size_t)n * n * sizeof(double));
            memcpy(Bb + (size_t)i * n * m, B, (size_t)n * m * sizeof(double));
        }
        double* xlo = (double*)malloc((size_t)n * sizeof(double));
        double* xhi = (double*)malloc((size_t)n * sizeof(double));
        for (int i = 0; i < n; ++i) { xlo[i] = -1e3; xhi[i] = 1e3; }
        CHECK(almpc_set_state_box(h, xlo, xhi));
        CHECK(almpc_design_batched(h, Ab, Bb, Q, R, NULL, NULL, 0, umin, umax, 0.1, 1e-6));
        CHECK(almpc_set_reference(h, xref, uref, 0));
        CHECK(almpc_update_initialization(h, x0));
        CHECK(almpc_calculate(h, &o));
        CHECK(almpc_get_results(h, x, ex, u, eu, status, NULL, NULL));
        for (int i = 0; i < batch; ++i) bad2 += status[i] != ALMPC_SOLVED;
        for (size_t t = 0; t < (size_t)batch * N * m; ++t) eu2_max = fmax(eu2_max, fabs(u[t] - ug[t]));
        almpc_destroy(h);
        free(Ab); free(Bb); free(xlo); free(xhi);
    }
    const int ok = bad == 0 && eu_max <= 1e-6 && ex_max <= 1e-5 && dev_max <= 1e-12 && P[0] > 0.0 &&
                   (comm_rc != ALMPC_OK || (summ[0] == 1 && summ[1] == 0)) && bad2 == 0 && eu2_max <= 1e-6;
    printf("%s instances %d unsolved %d max|u-u*| %.3e max|x-x*| %.3e deviation identity %.1e P11 %.6f comm_rc %d ranks %lld "
           "per-instance+state-box unsolved %d max|u-u*| %.3e\n",
           ok ? "OK" : "FAIL", batch, bad, eu_max, ex_max, dev_max, P[0], comm_rc, (long long)summ[0], bad2, eu2_max);
    return ok ? 0 : 1;
}
