// Driver of the CPU sanitizer build of the library's HOST half (round-4 review, item 9): csrc/almpc_api.hip compiled host-only under
// AddressSanitizer + UndefinedBehaviorSanitizer and linked against tests/sanitize/fake_hip_runtime.cpp (device memory = calloc'd host
// memory, launches = no-ops).  What runs is exactly the code a GPU lease cannot instrument: argument checks, buffer sizing, staging
// copies, launch-parameter set-up and every host-side readback of the C ABI, for a shared, a state-row, a per-instance, a structured,
// a re-linearised and an SQP handle, single and as a group, synchronous and through tickets.  The "results" are zeros (no kernel
// ran): the point is that no call touches memory it does not own.  Prints "host logic ok: <launches> launches".
#include "../../include/almpc.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" long fake_hip_launch_count();

#define CK(call)                                                                                             \
    do {                                                                                                     \
        const int rc_ = (call);                                                                              \
        if (rc_ != ALMPC_OK) { std::fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, h ? almpc_last_error(h) : ""); return 1; } \
    } while (0)
#define CKG(call)                                                                                            \
    do {                                                                                                     \
        const int rc_ = (call);                                                                              \
        if (rc_ != ALMPC_OK) { std::fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, almpc_group_last_error(g)); return 1; } \
    } while (0)

struct Plant {
    int n, m;
    std::vector<double> A, B, Q, R, S, umin, umax, xmin, xmax;
};

static Plant chain(int n, int m) {   // stable chain of integrators-with-leak, column-major
    Plant p;
    p.n = n; p.m = m;
    p.A.assign((size_t)n * n, 0.0); p.B.assign((size_t)n * m, 0.0); p.Q.assign((size_t)n * n, 0.0); p.R.assign((size_t)m * m, 0.0);
    p.S.assign((size_t)m * m, 0.0);
    for (int i = 0; i < n; ++i) {
        p.A[(size_t)i * n + i] = 0.9;
        if (i + 1 < n) p.A[(size_t)(i + 1) * n + i] = 0.1;
        p.Q[(size_t)i * n + i] = 100.0;
    }
    for (int a = 0; a < m; ++a) { p.B[(size_t)a * n + (n - 1 - a)] = 0.5; p.R[(size_t)a * m + a] = 0.1; p.S[(size_t)a * m + a] = 2.0; }
    p.umin.assign(m, -1.0); p.umax.assign(m, 1.0); p.xmin.assign(n, -3.0); p.xmax.assign(n, 3.0);
    return p;
}

static int step_and_read(almpc_handle* h, int n, int m, int N, int batch, bool tickets) {
    std::vector<double> x0((size_t)batch * n, 0.25), x((size_t)batch * n * (N + 1)), ex(x.size()), u((size_t)batch * m * N), eu(u.size()), u0((size_t)batch * m);
    std::vector<int32_t> st(batch), it(batch), pit(batch);
    almpc_opts o;
    almpc_default_opts(&o);
    CK(almpc_update_initialization(h, x0.data()));
    CK(almpc_calculate(h, &o));
    CK(almpc_get_results(h, x.data(), ex.data(), u.data(), eu.data(), st.data(), it.data(), pit.data()));
    CK(almpc_get_first_input(h, u0.data()));
    if (tickets) {
        double* slot = nullptr;
        CK(almpc_x0_staging(h, &slot));
        for (size_t i = 0; i < x0.size(); ++i) slot[i] = x0[i];
        CK(almpc_update_initialization_async(h, slot));
        CK(almpc_calculate_async(h, &o));
        const int t0 = almpc_get_results_async(h, ALMPC_WANT_FIRST_INPUT | ALMPC_WANT_STATUS);
        if (t0 < 0) return 1;
        CK(almpc_update_initialization_async(h, x0.data()));
        CK(almpc_calculate_async(h, &o));
        const int t1 = almpc_get_results_async(h, ALMPC_WANT_ALL);
        if (t1 < 0) return 1;
        CK(almpc_get_results_wait(h, t0, nullptr, nullptr, nullptr, nullptr, u0.data(), st.data(), nullptr, nullptr));
        CK(almpc_get_results_wait(h, t1, x.data(), ex.data(), u.data(), eu.data(), u0.data(), st.data(), it.data(), pit.data()));
        CK(almpc_synchronize(h));
    }
    return 0;
}

int main() {
    almpc_handle* h = nullptr;
    // ---- shared model, input box; then state box + terminal equality + S on the same handle; closed loop on the device
    {
        const int n = 4, m = 2, N = 12, batch = 37;
        const Plant p = chain(n, m);
        CK(almpc_create(&h, n, m, N, batch, 0, ALMPC_FLAG_TIMING));
        CK(almpc_set_rho_profile(h, 1));
        CK(almpc_design_shared(h, p.A.data(), p.B.data(), p.Q.data(), p.R.data(), nullptr, nullptr, p.umin.data(), p.umax.data(), nullptr, nullptr, 30.0, 1e-6));
        std::vector<double> xr((size_t)n * (N + 1), 0.1), ur((size_t)m * N, 0.05);
        for (int k = 0; k < N; ++k) ur[(size_t)k * m] = 0.01 * k;   // horizon-varying input reference
        CK(almpc_set_reference(h, xr.data(), ur.data(), 0));
        almpc_opts o;
        almpc_default_opts(&o);
        o.rho = 30.0;
        std::vector<double> x0((size_t)batch * n, 0.25);
        CK(almpc_update_initialization(h, x0.data()));
        CK(almpc_calculate(h, &o));
        CK(almpc_advance_plant(h));
        o.warm_start = 1;
        CK(almpc_calculate(h, &o));
        std::vector<double> H((size_t)m * N * m * N), F((size_t)m * N * n), P((size_t)n * n), d((size_t)m * N);
        CK(almpc_get_design(h, H.data(), F.data(), P.data(), d.data()));
        CK(almpc_set_terminal_equality(h, 1));
        CK(almpc_design_shared(h, p.A.data(), p.B.data(), p.Q.data(), p.R.data(), p.S.data(), nullptr, p.umin.data(), p.umax.data(), p.xmin.data(), p.xmax.data(), 0.1, 1e-6));
        std::vector<double> xrb((size_t)batch * n * (N + 1), 0.0), urb((size_t)batch * m * N, 0.0);
        CK(almpc_set_reference(h, xrb.data(), urb.data(), 1));   // per-instance references
        if (step_and_read(h, n, m, N, batch, true)) return 1;
        almpc_destroy(h); h = nullptr;
    }
    // ---- the benchmark shape (fused one-kernel step, nz = 120) and the per-instance-model path on it
    {
        const int n = 12, m = 4, N = 30, batch = 48;
        const Plant p = chain(n, m);
        CK(almpc_create(&h, n, m, N, batch, 0, 0));
        CK(almpc_design_shared(h, p.A.data(), p.B.data(), p.Q.data(), p.R.data(), nullptr, nullptr, p.umin.data(), p.umax.data(), nullptr, nullptr, 0.1, 1e-6));
        std::vector<double> xz((size_t)n * (N + 1), 0.0), uz((size_t)m * N, 0.0);
        CK(almpc_set_reference(h, xz.data(), uz.data(), 0));
        if (step_and_read(h, n, m, N, batch, true)) return 1;
        std::vector<double> Ab((size_t)batch * n * n), Bb((size_t)batch * n * m);
        for (int i = 0; i < batch; ++i) {
            for (size_t t = 0; t < p.A.size(); ++t) Ab[(size_t)i * n * n + t] = p.A[t];
            for (size_t t = 0; t < p.B.size(); ++t) Bb[(size_t)i * n * m + t] = p.B[t] * (1.0 + 0.01 * i);
        }
        CK(almpc_set_state_box(h, p.xmin.data(), p.xmax.data()));
        CK(almpc_design_batched(h, Ab.data(), Bb.data(), p.Q.data(), p.R.data(), nullptr, nullptr, 0, p.umin.data(), p.umax.data(), 0.1, 1e-6));
        CK(almpc_set_reference(h, xz.data(), uz.data(), 0));
        if (step_and_read(h, n, m, N, batch, false)) return 1;
        std::vector<double> Hi((size_t)m * N * m * N), Fi((size_t)m * N * n), di((size_t)m * N);
        CK(almpc_get_design_instance(h, batch - 1, Hi.data(), Fi.data(), di.data()));
        almpc_destroy(h); h = nullptr;
    }
    // ---- structured handle beyond the condensed horizon (m N = 200), state box + S
    {
        const int n = 12, m = 4, N = 50, batch = 21;
        const Plant p = chain(n, m);
        CK(almpc_create(&h, n, m, N, batch, 0, ALMPC_FLAG_STRUCTURED));
        CK(almpc_design_shared(h, p.A.data(), p.B.data(), p.Q.data(), p.R.data(), p.S.data(), nullptr, p.umin.data(), p.umax.data(), p.xmin.data(), p.xmax.data(), 0.1, 1e-6));
        std::vector<double> xz((size_t)n * (N + 1), 0.0), uz((size_t)m * N, 0.0);
        CK(almpc_set_reference(h, xz.data(), uz.data(), 0));
        if (step_and_read(h, n, m, N, batch, true)) return 1;
        almpc_destroy(h); h = nullptr;
    }
    // ---- re-linearisation pipeline and SQP loop on a small network
    {
        const int n = 4, m = 2, N = 20, batch = 19, Hn = 8, L = 2;
        const Plant p = chain(n, m);
        std::vector<double> W_in((size_t)Hn * (n + m), 0.05), W_h((size_t)L * Hn * Hn, 0.02), b_h((size_t)L * Hn, 0.01), W_out((size_t)n * Hn, 0.1);
        std::vector<double> P((size_t)n * n, 0.0), xr((size_t)n * (N + 1), 0.0), ur((size_t)m * N, 0.0);
        for (int i = 0; i < n; ++i) P[(size_t)i * n + i] = 150.0;
        for (int k = 0; k < N; ++k) ur[(size_t)k * m + 1] = 0.02 * k;
        CK(almpc_create(&h, n, m, N, batch, 0, ALMPC_FLAG_TIMING));
        CK(almpc_relin_fnn_setup(h, Hn, L, 0, W_in.data(), W_h.data(), b_h.data(), W_out.data(), xr.data(), ur.data(), p.Q.data(), p.R.data(), p.S.data(),
                                 P.data(), p.umin.data(), p.umax.data(), 0.1, 1e-6));
        std::vector<double> x0((size_t)batch * n, 0.1);
        CK(almpc_update_initialization(h, x0.data()));
        almpc_opts o;
        almpc_default_opts(&o);
        CK(almpc_relin_fnn_step(h, &o));
        CK(almpc_relin_fnn_advance(h));
        o.warm_start = 1;
        CK(almpc_relin_fnn_step(h, &o));
        float a, b, c;
        CK(almpc_relin_fnn_timing(h, &a, &b, &c));
        CK(almpc_sqp_fnn_setup(h, Hn, L, 1, W_in.data(), W_h.data(), b_h.data(), W_out.data(), xr.data(), ur.data(), p.Q.data(), p.R.data(), nullptr,
                               P.data(), 0, p.umin.data(), p.umax.data(), 0.1, 1e-6));
        CK(almpc_sqp_fnn_start(h, x0.data(), nullptr));
        std::vector<double> si(3), di(3);
        const int rc = almpc_sqp_fnn_iterate(h, 3, 1.0, nullptr, si.data(), di.data());
        if (rc != ALMPC_OK && rc != ALMPC_ERR_NUMERIC) { std::fprintf(stderr, "sqp iterate -> %d (%s)\n", rc, almpc_last_error(h)); return 1; }
        std::vector<int32_t> sk(batch);
        CK(almpc_sqp_fnn_skipped(h, sk.data()));
        almpc_destroy(h); h = nullptr;
    }
    // ---- a group of three handles (all on the one fake device): uneven shards, tickets
    {
        almpc_group* g = nullptr;
        const int n = 4, m = 2, N = 10, batch = 50, ids[3] = {0, 0, 0};
        const Plant p = chain(n, m);
        CKG(almpc_group_create(&g, n, m, N, batch, 3, ids, 0));
        CKG(almpc_group_design_shared(g, p.A.data(), p.B.data(), p.Q.data(), p.R.data(), nullptr, nullptr, p.umin.data(), p.umax.data(), nullptr, nullptr, 0.1, 1e-6));
        std::vector<double> xz((size_t)n * (N + 1), 0.0), uz((size_t)m * N, 0.0);
        CKG(almpc_group_set_reference(g, xz.data(), uz.data(), 0));
        std::vector<double> x0((size_t)batch * n, 0.2), x((size_t)batch * n * (N + 1)), u((size_t)batch * m * N), u0((size_t)batch * m);
        std::vector<int32_t> st(batch);
        CKG(almpc_group_update_initialization(g, x0.data()));
        CKG(almpc_group_calculate(g, nullptr));
        CKG(almpc_group_get_results(g, x.data(), nullptr, u.data(), nullptr, u0.data(), st.data(), nullptr, nullptr));
        CKG(almpc_group_calculate_async(g, nullptr));
        const int t = almpc_group_get_results_async(g, ALMPC_WANT_FIRST_INPUT | ALMPC_WANT_STATUS | ALMPC_WANT_X);
        if (t < 0) return 1;
        CKG(almpc_group_get_results_wait(g, t, x.data(), nullptr, nullptr, nullptr, u0.data(), st.data(), nullptr, nullptr));
        CKG(almpc_group_advance_plant(g));
        almpc_group_destroy(g);
    }
    std::printf("host logic ok: %ld launches\n", fake_hip_launch_count());
    return 0;
}
