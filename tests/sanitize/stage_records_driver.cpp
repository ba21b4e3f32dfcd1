// Sanitizer driver for hm::stage_records (csrc/almpc_host_math.h: the host Riccati recursion behind k_sdual's stage records): compiled by
// tests/test_stagewise_oracle.py with -fsanitize=address,undefined; reads n m N useS NT MC, A, B, Q, R, S, P; prints `ok invariant`
// and the records.  Test infrastructure only.
#include "../../automationlabsmodelpredictivecontrol.jl_amd/csrc/almpc_host_math.h"

#include <cstdio>
#include <cstdlib>
using namespace almpc::hm;

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = std::fopen(argv[1], "r");
    if (!f) return 2;
    int n, m, N, useS, NT, MC;
    if (std::fscanf(f, "%d %d %d %d %d %d", &n, &m, &N, &useS, &NT, &MC) != 6) return 2;
    auto rd = [&](size_t cnt) { mat v(cnt); for (auto& x : v) if (std::fscanf(f, "%lf", &x) != 1) std::exit(2); return v; };
    mat A = rd((size_t)n * n), B = rd((size_t)n * m), Q = rd((size_t)n * n), R = rd((size_t)m * m), S = rd((size_t)m * m), P = rd((size_t)n * n), rec;
    std::fclose(f);
    bool inv = false;
    const bool ok = stage_records(A, B, Q, R, useS ? &S : nullptr, P, n, m, N, NT, MC, rec, inv);
    std::printf("%d %d\n", ok ? 1 : 0, inv ? 1 : 0);
    if (ok) for (double v : rec) std::printf("%.17g\n", v);
    return 0;
}
