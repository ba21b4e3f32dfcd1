/*
 * almpc_oracle.c -- CPU restatement (plain C, float64) of the per-step MPC solve.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing here is linked into, loaded by or called from the product
 * (libalmpc.so / the Python host package).  Users: tests/, __graft_entry__.smoke() as the checker,
 * and bench.py's `cpu_baseline` leg (kind "port": the reference itself -- Julia + OSQP -- cannot
 * run in this image).
 *
 * What it follows (reference paths relative to /root/reference):
 *   per-step I/O            src/main/computation_mpc.jl:17-55  (x0 in; u, e_u, x, e_x out)
 *   QP being solved         src/sub/model_modeler_implementation/linear/mpc_modeler_implementation_linear.jl:48-100,
 *                           src/sub/design_mpc.jl:405-468   (condensed as in SURVEY.md section 8a; the design
 *                           matrices are produced by oracle/mpc_oracle.py::design_shared and passed in)
 *   solver arithmetic       OSQP (third-party, not in /root/reference; Julia pkg OSQP compat "0.8",
 *                           Project.toml:28): Stellato et al., "OSQP: an operator splitting solver for
 *                           quadratic programs", Algorithm 1, box specialisation A = I, followed by an
 *                           active-set polish -- the same mathematics as mpc_oracle.py::admm_box and
 *                           ::polish_active_set (which tests/test_oracle.py pins this file against).
 *
 * Parity pin status: see the header of oracle/mpc_oracle.py ("parity unpinned" beyond the reference's
 * own loose assertions; truth for the 1e-5 bar is the KKT-certified exact optimum).
 *
 * All matrices column-major; symmetric nz x nz matrices have leading dimension nz.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double clip(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* y = M x (M symmetric nz x nz, column-major) */
static void symv(int nz, const double* M, const double* x, double* y) {
    for (int i = 0; i < nz; ++i) y[i] = 0.0;
    for (int j = 0; j < nz; ++j) {
        const double xj = x[j];
        const double* col = M + (size_t)j * nz;
        for (int i = 0; i < nz; ++i) y[i] += col[i] * xj;
    }
}

/*
 * Box ADMM in scaled coordinates (mpc_oracle.py::admm_box).  Returns status 0 solved / 1 max-iter /
 * 2 non-finite.  x, z, y: in = start (cold start: zeros), out = final iterate.  work: 4*nz doubles.
 */
int almpc_oracle_admm(int nz, const double* Minv, const double* Hs, const double* fs, const double* lo,
                      const double* hi, const double* d, const double* rho, double sigma, double alpha, double eps_abs,
                      double eps_rel, int max_iter, int check_every, double* x, double* z, double* y, int* iters,
                      double* work) {
    double* rhs = work;
    double* xt = work + nz;
    double* Hx = work + 2 * nz;
    int status = 1, it = 0;
    for (int i = 0; i < nz; ++i) z[i] = clip(z[i], lo[i], hi[i]);
    for (it = 1; it <= max_iter; ++it) {
        for (int i = 0; i < nz; ++i) rhs[i] = sigma * x[i] - fs[i] + rho[i] * z[i] - y[i];
        symv(nz, Minv, rhs, xt);
        for (int i = 0; i < nz; ++i) {
            x[i] = alpha * xt[i] + (1.0 - alpha) * x[i];
            const double w = alpha * xt[i] + (1.0 - alpha) * z[i] + y[i] / rho[i];
            const double zn = clip(w, lo[i], hi[i]);
            y[i] = rho[i] * (w - zn);
            z[i] = zn;
        }
        if (it % check_every == 0 || it == max_iter) {
            symv(nz, Hs, x, Hx);
            double rp = 0, rd = 0, nx = 0, nzn = 0, nhx = 0, ny = 0, nf = 0;
            int bad = 0;
            for (int i = 0; i < nz; ++i) {
                rp = fmax(rp, fabs(d[i] * (x[i] - z[i])));
                nx = fmax(nx, fabs(d[i] * x[i]));
                nzn = fmax(nzn, fabs(d[i] * z[i]));
                rd = fmax(rd, fabs((Hx[i] + fs[i] + y[i]) / d[i]));
                nhx = fmax(nhx, fabs(Hx[i] / d[i]));
                ny = fmax(ny, fabs(y[i] / d[i]));
                nf = fmax(nf, fabs(fs[i] / d[i]));
                if (!isfinite(x[i] + y[i] + Hx[i])) bad = 1;
            }
            if (bad) { status = 2; break; }
            if (rp <= eps_abs + eps_rel * fmax(nx, nzn) && rd <= eps_abs + eps_rel * fmax(fmax(nhx, ny), nf)) {
                status = 0;
                break;
            }
        }
    }
    if (it > max_iter) it = max_iter;
    *iters = it;
    return status;
}

/*
 * Active-set polish with the shared inverse G = Hs^-1 (mpc_oracle.py::polish_active_set, refine=False).
 * v0 = -G fs.  In: z, y (ADMM iterate).  Out: w.  Returns 0 if the multipliers certify optimality,
 * 1 if max_iter was hit.  work: nz*nz + 8*nz doubles, iwork: 3*nz ints.
 */
int almpc_oracle_polish(int nz, const double* G, const double* v0, const double* lo, const double* hi,
                        const double* z, const double* y, int max_iter, double* w, int* iters, double* work,
                        int* iwork) {
    double* S = work;               /* (G_WW)^-1, k x k, leading dimension nz */
    double* c = S + (size_t)nz * nz;
    double* u = c + nz;
    double* lam = u + nz;
    double* r = lam + nz;
    double* t = r + nz;
    double* bnd = t + nz;           /* bound value per working-set position */
    int* W = iwork;                 /* position -> row */
    int* side = iwork + nz;         /* position -> +1 / -1 */
    int* pos = iwork + 2 * nz;      /* row -> position or -1 */
    int k = 0, it = 0, fin = 1;
    for (int i = 0; i < nz; ++i) { pos[i] = -1; w[i] = clip(z[i], lo[i], hi[i]); }

#define ADD(J, SD, BV)                                                                     \
    do {                                                                                   \
        const int j_ = (J);                                                                \
        for (int i = 0; i < k; ++i) c[i] = G[(size_t)j_ * nz + W[i]];                      \
        for (int i = 0; i < k; ++i) {                                                      \
            double s_ = 0.0;                                                               \
            for (int l = 0; l < k; ++l) s_ += S[(size_t)l * nz + i] * c[l];                \
            u[i] = s_;                                                                     \
        }                                                                                  \
        double sc_ = G[(size_t)j_ * nz + j_];                                              \
        for (int i = 0; i < k; ++i) sc_ -= c[i] * u[i];                                    \
        const double isc_ = 1.0 / sc_;                                                     \
        for (int l = 0; l < k; ++l)                                                        \
            for (int i = 0; i < k; ++i) S[(size_t)l * nz + i] += u[i] * u[l] * isc_;       \
        for (int i = 0; i < k; ++i) { S[(size_t)k * nz + i] = -u[i] * isc_; S[(size_t)i * nz + k] = -u[i] * isc_; } \
        S[(size_t)k * nz + k] = isc_;                                                      \
        W[k] = j_; side[k] = (SD); bnd[k] = (BV); pos[j_] = k; ++k;                        \
    } while (0)

    for (int j = 0; j < nz; ++j) {
        if (y[j] < 0.0 && w[j] <= lo[j]) ADD(j, -1, lo[j]);
        else if (y[j] > 0.0 && w[j] >= hi[j]) ADD(j, +1, hi[j]);
    }
    /* One-shot purge (as in k_polish): if the multipliers of a guessed set of at most 32 rows come out with the wrong
     * sign on two or more rows, those rows are dropped together and the inverse is rebuilt for the others. */
    if (k > 0 && k <= 32) {
        for (int i = 0; i < k; ++i) r[i] = v0[W[i]] - bnd[i];
        double lmax = 0.0;
        for (int i = 0; i < k; ++i) {
            double s = 0.0;
            for (int l = 0; l < k; ++l) s += S[(size_t)l * nz + i] * r[l];
            lam[i] = s;
            lmax = fmax(lmax, fabs(s));
        }
        int nb = 0;
        for (int i = 0; i < k; ++i) nb += ((side[i] > 0 ? -lam[i] : lam[i]) > 1e-12 * fmax(1.0, lmax)) ? 1 : 0;
        if (nb >= 2) {
            int kr[32], ks[32], nk = 0;
            double kb[32];
            for (int i = 0; i < k; ++i) {
                if ((side[i] > 0 ? -lam[i] : lam[i]) > 1e-12 * fmax(1.0, lmax)) pos[W[i]] = -1;
                else { kr[nk] = W[i]; ks[nk] = side[i]; kb[nk] = bnd[i]; ++nk; }
            }
            k = 0;
            for (int i = 0; i < nk; ++i) ADD(kr[i], ks[i], kb[i]);
            it += nb;
        }
    }
    while (it < max_iter) {
        ++it;
        for (int i = 0; i < k; ++i) r[i] = v0[W[i]] - bnd[i];
        for (int i = 0; i < k; ++i) {
            double s = 0.0;
            for (int l = 0; l < k; ++l) s += S[(size_t)l * nz + i] * r[l];
            lam[i] = s;
        }
        for (int i = 0; i < nz; ++i) t[i] = v0[i];
        for (int l = 0; l < k; ++l) {
            const double* col = G + (size_t)W[l] * nz;
            const double ll = lam[l];
            for (int i = 0; i < nz; ++i) t[i] -= col[i] * ll;
        }
        for (int i = 0; i < k; ++i) t[W[i]] = bnd[i];
        /* ratio test over the free rows; ties -> smallest row */
        double rmin = INFINITY;
        int jmin = -1, smin = 0;
        for (int i = 0; i < nz; ++i) {
            if (pos[i] >= 0) continue;
            double rr = INFINITY;
            int s = 0;
            if (t[i] > hi[i]) { rr = (hi[i] - w[i]) / (t[i] - w[i]); s = +1; }
            else if (t[i] < lo[i]) { rr = (lo[i] - w[i]) / (t[i] - w[i]); s = -1; }
            if (rr < rmin) { rmin = rr; jmin = i; smin = s; }
        }
        if (rmin < 1.0) {
            const double tt = rmin > 0.0 ? rmin : 0.0;
            for (int i = 0; i < nz; ++i)
                if (pos[i] < 0) w[i] += tt * (t[i] - w[i]);
            const double bv = smin > 0 ? hi[jmin] : lo[jmin];
            w[jmin] = bv;
            ADD(jmin, smin, bv);
            continue;
        }
        for (int i = 0; i < nz; ++i) w[i] = t[i];
        if (k == 0) { fin = 0; break; }
        double vmax = -INFINITY, lmax = 0.0;
        int vi = -1;
        for (int i = 0; i < k; ++i) {
            const double v = side[i] > 0 ? -lam[i] : lam[i];
            if (v > vmax) { vmax = v; vi = i; }
            lmax = fmax(lmax, fabs(lam[i]));
        }
        if (vmax <= 1e-12 * fmax(1.0, lmax)) { fin = 0; break; }
        /* remove position vi: Schur down-date, then move the last row/column into the hole */
        {
            const double ipp = 1.0 / S[(size_t)vi * nz + vi];
            for (int i = 0; i < k; ++i) c[i] = S[(size_t)vi * nz + i];
            for (int l = 0; l < k; ++l)
                for (int i = 0; i < k; ++i) S[(size_t)l * nz + i] -= c[i] * c[l] * ipp;
            const int last = k - 1;
            pos[W[vi]] = -1;
            if (vi != last) {
                for (int i = 0; i < k; ++i) c[i] = S[(size_t)last * nz + i];
                for (int i = 0; i < k; ++i)
                    if (i != vi && i != last) { S[(size_t)vi * nz + i] = c[i]; S[(size_t)i * nz + vi] = c[i]; }
                S[(size_t)vi * nz + vi] = c[last];
                W[vi] = W[last]; side[vi] = side[last]; bnd[vi] = bnd[last]; pos[W[vi]] = vi;
            }
            --k;
        }
    }
#undef ADD
    for (int i = 0; i < nz; ++i) w[i] = clip(w[i], lo[i], hi[i]);
    *iters = it;
    return fin;
}

/* outputs of calculate! for one instance: u, e_u [N][m]; x, e_x [N+1][n] (column-major n x (N+1)) */
void almpc_oracle_rollout(int n, int m, int N, const double* A, const double* B, const double* xref,
                          const double* uref, const double* x0, const double* v, double* x, double* e_x, double* u,
                          double* e_u) {
    for (int r = 0; r < m * N; ++r) { e_u[r] = v[r]; u[r] = v[r] + uref[r]; }
    for (int i = 0; i < n; ++i) { e_x[i] = x0[i] - xref[i]; x[i] = x0[i]; }
    for (int k = 0; k < N; ++k) {
        const double* e = e_x + (size_t)k * n;
        double* en = e_x + (size_t)(k + 1) * n;
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += A[(size_t)j * n + i] * e[j];
            for (int j = 0; j < m; ++j) s += B[(size_t)j * n + i] * v[k * m + j];
            en[i] = s;
            x[(size_t)(k + 1) * n + i] = s + xref[(size_t)(k + 1) * n + i];
        }
    }
}

/*
 * One MPC step for a batch (shared model, shared references): gradient, ADMM, polish, rollout.
 * Fs: nz x n column-major (scaled), fS: nz (scaled), lo/hi: nz (scaled), x0: [batch][n].
 * Outputs: u, e_u [batch][N][m]; x, e_x [batch][N+1][n]; status, iters, piters [batch].
 * threads <= 0: all cores.  Returns the number of threads used, or -1 on allocation failure.
 */
int almpc_oracle_step_batch(int n, int m, int N, int batch, const double* A, const double* B, const double* Minv,
                            const double* Hs, const double* G, const double* Fs, const double* fS, const double* lo,
                            const double* hi, const double* d, const double* xref, const double* uref,
                            const double* x0, const double* rho, double sigma, double alpha, double eps_abs, double eps_rel,
                            int max_iter, int check_every, int polish, int polish_max_iter, double* x, double* e_x,
                            double* u, double* e_u, int32_t* status, int32_t* iters, int32_t* piters, int threads) {
    const int nz = m * N;
    int used = 1, failed = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    used = threads;
#else
    (void)threads;
#endif
    if (polish_max_iter <= 0) polish_max_iter = 2 * nz + 50;
#pragma omp parallel num_threads(used)
    {
        double* work = (double*)malloc(sizeof(double) * ((size_t)nz * nz + 20 * (size_t)nz + n));
        int* iwork = (int*)malloc(sizeof(int) * 3 * (size_t)nz);
        if (!work || !iwork) {
#pragma omp atomic write
            failed = 1;
        } else {
            double* fs = work;
            double* xs = fs + nz;
            double* zs = xs + nz;
            double* ys = zs + nz;
            double* v0 = ys + nz;
            double* w = v0 + nz;
            double* v = w + nz;
            double* e0 = v + nz;
            double* rest = e0 + n; /* >= nz*nz + 8*nz */
#pragma omp for schedule(dynamic, 8)
            for (int b = 0; b < batch; ++b) {
                for (int i = 0; i < n; ++i) e0[i] = x0[(size_t)b * n + i] - xref[i];
                for (int r = 0; r < nz; ++r) fs[r] = fS[r];
                for (int j = 0; j < n; ++j) {
                    const double* col = Fs + (size_t)j * nz;
                    for (int r = 0; r < nz; ++r) fs[r] += col[r] * e0[j];
                }
                memset(xs, 0, sizeof(double) * 3 * (size_t)nz);
                int it = 0, pit = 0;
                int st = almpc_oracle_admm(nz, Minv, Hs, fs, lo, hi, d, rho, sigma, alpha, eps_abs, eps_rel, max_iter,
                                           check_every, xs, zs, ys, &it, rest);
                memcpy(w, zs, sizeof(double) * nz);
                if (polish && st != 2) {
                    for (int r = 0; r < nz; ++r) rest[r] = -fs[r];
                    symv(nz, G, rest, v0);
                    const int fin = almpc_oracle_polish(nz, G, v0, lo, hi, zs, ys, polish_max_iter, w, &pit, rest, iwork);
                    if (fin == 0) st = 0;
                }
                for (int r = 0; r < nz; ++r) v[r] = w[r] * d[r];
                almpc_oracle_rollout(n, m, N, A, B, xref, uref, x0 + (size_t)b * n, v, x + (size_t)b * n * (N + 1),
                                     e_x + (size_t)b * n * (N + 1), u + (size_t)b * nz, e_u + (size_t)b * nz);
                status[b] = st; iters[b] = it; piters[b] = pit;
            }
        }
        free(work);
        free(iwork);
    }
    return failed ? -1 : used;
}
