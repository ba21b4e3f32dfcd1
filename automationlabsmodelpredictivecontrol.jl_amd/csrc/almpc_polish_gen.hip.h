// almpc_polish_gen.hip.h -- exact finish for problems with STATE rows (state box, terminal equality).
//
// Reference rows restated (paths relative to /root/reference):
//   state box  x_min <= x[i,k] <= x_max, k = 1..N+1, only when kw `mpc_state_constraint` is present
//              src/sub/model_modeler_implementation/linear/mpc_modeler_implementation_linear.jl:62-70
//   terminal equality  e_x[:,N+1] == 0      src/sub/design_mpc.jl:330-331
//
// Method: Goldfarb-Idnani dual active set in CONSTRAINT space (oracle: mpc_oracle.py::solve_qp_dual_active_set).
// Rows of A = [I; C'] (input box rows, then state rows; C' = rows of Gamma D), shared Ghat = A H'^-1 A' (R x R).
// With working set W at bounds b:  s = s0 - Ghat[:,W] lam,  Ghat[W,W] lam = s0_W - b,  s0 = A v0 + [0; Phi e0]
// (= the unconstrained minimiser v0 and ITS state trajectory, which one rollout gives).  The box-only ADMM of
// k_admm supplies v0 and the guess for the input rows; violated rows (input or state) are then added one at a
// time, with partial steps and drops when a multiplier would change sign.  Dual feasible throughout, so the first
// primal-feasible iterate is the optimum; infeasibility shows as a dependent violated row without a blocking
// multiplier (status ALMPC_INFEASIBLE).
//
// One wave per instance; lane l owns row pairs (2l, 2l+1) + 128 q, q < NP (R <= 128 NP).  Same machinery as
// k_polish: Sinv = (Ghat_WW)^-1 in LDS (WL x WL, identity padded), position-distributed data in registers, LDS broadcasts,
// DPP reductions, branch-free sweeps.  Two builds: WL = 32 (both half-waves mirror the 32 positions and split the columns of
// every sweep; 17 KB of LDS per single-wave workgroup) runs on every instance; an instance whose working set outgrows 32 rows is
// flagged and continued by the WL = 64 build (one position per lane, 42 KB per wave) in a second launch that every other wave leaves
// at once -- from the first launch's working set AND inverse (ovf_ws, ovf_sinv).
#pragma once
#include "almpc_kernels.hip.h"

namespace almpc {

struct PolishGenParams {
    int nz, mc, R, Rs, m, n, N, batch, nzs;
    const double* Ghat;   // [R][Rs] dense symmetric
    const double* gnorm;  // [Rs] sqrt(Ghat_rr) (pad rows: 1)
    long Ghat_stride = 0, gnorm_stride = 0, d_stride = 0, A_stride = 0, B_stride = 0;  // per-instance models: doubles between instances
    // time-varying designs (almpc_design_ltv, the SQP loop): the state rows are rows of dx, dx_{k+1} = A_k dx_k + B_k v_k + c_k, dx_0 = 0,
    // with bounds relative to the linearisation trajectory (which sits in the reference slots); no x / e_x outputs
    int ltv = 0;
    const double* ltvA = nullptr;  // [batch][N][n*n]
    const double* ltvB = nullptr;  // [batch][N][n*m]
    const double* ltvC = nullptr;  // [batch][N][n] or null
    const double* eq_off = nullptr;  // [batch][eq_stride]: terminal equality e_x[:,N+1] = 0 in dx: dx_N = -eq_off (= x_ref - xbar at N+1)
    long eq_stride = 0;
    // Shared-model designs with the terminal equality: the ne equality rows (rows eq0 .. eq0 + ne - 1, always in the working set) are
    // eliminated at design time -- Ghat is the PROJECTED matrix Ghat - Ghat[:,E] Ghat_EE^-1 Ghat[E,:] (zero rows / columns E), and
    // every instance starts from s0 - Ghat[:,E] Ghat_EE^-1 (s0_E - b): the minimiser subject to the equality.  The finish then runs
    // on the remaining rows only: 12 bordering steps per instance (66 k cycles) and 12 of the 32 working-set slots are saved.
    // shared model with the design's blocked-rollout table (rollout_blocked, csrc/almpc_kernels.hip.h): the output rollout advances s
    // stages at a time and streams x / e_x to HBM (14.9 k + 4.5 k cycles -> 10.7 k + 0.6 k at the quadrotor shape); null: stage by
    // stage.  (The v0 trajectory of the prologue stays stage by stage: its values are needed in LDS, and the blocked form through a
    // generic pointer plus 80 more live registers cost more there than the shorter chain saved -- measured.)
    const double* rollM = nullptr;
    int roll_s = 0, roll_nb = 0;
    int eq_proj = 0, eq0 = 0, ne = 0;
    const double* GhatE = nullptr;   // [ne][Rs] the ORIGINAL rows E of Ghat
    const double* WinvE = nullptr;   // [ne][ne] Ghat_EE^-1
    const int* row_traj;  // [Rs] for state rows: offset of e_x[i,k] in the wave's trajectory buffer Z ((k)*C + i); else -1
    const int* row_eq;    // [Rs] 1 for terminal-equality rows
    const int* row_xidx;  // [Rs] for state rows: i + n*k index into x_ref (k = reference stage 2..N+1 -> 1..N); else 0
    const int* row_state; // [Rs] for state rows: state index i; else 0
    const double* xmin;   // [n] (only read when has_box)
    const double* xmax;
    int has_box;
    const double* dvec; const double* umin; const double* umax;
    const double* uref; long uref_stride;
    const double* zs; const double* ys; const double* v0;
    int32_t* status; int32_t* piters;
    int32_t* ovf;         // [0] number of flagged instances (working set outgrew the 32-row build), [1] queue cursor of the
                          // second launch, [2 ...] the flagged instances; [0] and [1] are zeroed before the first launch
    int32_t* ovf_ws;      // [batch][32] working set of a flagged instance when it ran out of room, row | (side + 1) << 12, the
                          // second launch's guess ([0] = -1: none, it builds its guess from the ADMM hand-off as the first did)
    // Shared model AND shared references: the row values of the unconstrained minimiser are affine in e0 -- s0_state = M1 e0 + m0 with
    // M1 = C'V + Phi, m0 = C' v0S (the v0 rollout of the prologue, 15 - 17 k cycles on one wave, becomes n + 1 coalesced rows of a
    // design-time table).  s0_basis: [(n + 1)][Rs], row j < n = the state-row values for e0 = unit_j (v0 = V[:, j]), row n = for e0 = 0
    // (v0 = v0S); built by k_s0_basis with this kernel's own rollout.  Null: roll out.
    const double* s0_basis = nullptr;
    double* ovf_sinv = nullptr;  // [batch][32 * 32 + 32] of a flagged instance: the 32 x 32 inverse and the positions' bounds at that moment, so
                                 // that the second launch continues instead of bordering the 32 rows in again (105 k cycles), or null
    int* unsolved = nullptr;   // host-visible count of instances left with ALMPC_MAX_ITER (lazy redo, see almpc_handle::hUnsolved), or null
    int* redo_gate = nullptr; int step_serial = 0;   // as PolishParams::redo_gate
    // start of the stage-wise redo (round 5): an instance this finish leaves undecided hands k_sdual its working set, in the coordinates
    // of the stage-wise form -- [batch][64]: count, then (coordinate << 2 | 1 upper / 0 lower / 2 equality) -- or null.  The rows are then
    // bordered in one cached response each instead of being found again one scan / direction / ratio test at a time.
    int32_t* redo_ws = nullptr;
    int redo_sp = 0, redo_nt = 0;   // k_sdual's stage layout: coordinates per stage, offset of the input slot
    int max_iter;
    int roll_g, roll_cpl;
    RolloutParams roll;
};

// one instance per workgroup of the first launch: a CU's eight slots are refilled instance by instance (with four per workgroup a slot
// waited for the slowest of four: tight box 550 -> 487 us for the finish of 4096 instances)
constexpr int PGEN_WAVES = 1;
constexpr int PGEN_HELPERS = 1;   // helper waves of the second launch's workgroup (see k_polish_gen64)
// LDS per wave (doubles): Sinv WL x WL | rowbuf Rs(<=512) | s0buf Rs(<=512) | pbufa WL | pbufb WL | wrow_s (WL ints) | Z trajectory shares Sinv
__host__ __device__ constexpr int pgen_lds_per_wave(int WL) { return WL * WL + 512 + 512 + WL + WL + WL / 2; }

// Rollout through stage models (time-varying designs): Z rows [dx(k); v(k)], dx(0) in Z[0..n); lane layout as rollout_steps, the
// stage's coefficients come from global memory and are requested one stage ahead of the chain.
template <int CPL>
__device__ __forceinline__ void rollout_steps_ltv(double* Z, int n, int m, int N, int G, int lane, const double* A, const double* B,
                                                  const double* c) {
    const int C = n + m;
    const int i = lane / G, g = lane % G, ic = i < n ? i : 0;
    int jc[CPL];
    size_t offA[CPL], offB[CPL];
    bool isA[CPL], ok[CPL];
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
        const int j = g * CPL + t;
        ok[t] = (i < n) && (j < C);
        isA[t] = j < n;
        offA[t] = (size_t)(j < n ? j : 0) * n + ic;
        offB[t] = (size_t)((j >= n && j < C) ? j - n : 0) * n + ic;
        jc[t] = j < C ? j : C - 1;
    }
    auto load = [&](int k, double (&cf)[CPL], double& ck) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            const double av = A[(size_t)k * n * n + offA[t]], bv = B[(size_t)k * n * m + offB[t]];
            cf[t] = ok[t] ? (isA[t] ? av : bv) : 0.0;
        }
        ck = c ? c[(size_t)k * n + ic] : 0.0;
    };
    double cf[CPL], ck;
    load(0, cf, ck);
    for (int k = 0; k < N; ++k) {
        double nf[CPL], nck = 0.0;
        load(k + 1 < N ? k + 1 : k, nf, nck);
        const double* zk = Z + (size_t)k * C;
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < CPL; ++t) acc += cf[t] * zk[jc[t]];
        for (int o = 1; o < G; o <<= 1) acc += __shfl_xor(acc, o);
        if (g == 0 && i < n) Z[(size_t)(k + 1) * C + i] = acc + ck;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int t = 0; t < CPL; ++t) cf[t] = nf[t];
        ck = nck;
    }
}

// COOP (second launch): the workgroup's other three waves take shares of the Ghat[W,:] stream (pgen_helper below)
template <int NP, int WL, bool COOP = false>
__device__ __forceinline__ void polish_gen_body(const PolishGenParams& p, double* smem) {
    static_assert(WL == 32 || WL == 64, "working-set capacity: 32 (mirrored half-waves) or 64 (one position per lane)");
    constexpr int HS = 64 / WL;  // lanes per position: 2 = the half-waves split the columns of a sweep, 1 = no split
    constexpr int CH = (NP >= 3 && WL == 32) ? 4 : 8;  // rows of Ghat per group of loads: NP * CH <= 16 loads (64 registers) in flight per lane; 32 in the second launch (512 registers)
    constexpr bool QUEUE = WL == 64;  // the 64-row build is the second launch: persistent waves pull flagged instances
    const int wv = COOP ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane0 = threadIdx.x & 63;   // (wave-uniform, and the compiler knows it)
  do {  // (one pass in the first launch)
    // (second launch: the lane index made opaque per instance -- otherwise everything that depends on it alone is hoisted out of this
    // loop and kept live across the whole solve: 650 live registers, 134 of them in scratch memory)
    int lane = lane0;
    if (QUEUE) asm volatile("" : "+v"(lane));
    int inst;
    if (QUEUE) {
        if (p.ovf[0] == 0) return;  // (nothing flagged: leave without touching the cursor)
        int qi = 0;
        if (lane == 0) qi = atomicAdd(p.ovf + 1, 1);
        qi = __builtin_amdgcn_readfirstlane(qi);
        if (qi >= p.ovf[0]) return;
        inst = p.ovf[2 + qi];
    } else {
        inst = blockIdx.x * PGEN_WAVES + wv;
        if (inst >= p.batch) return;
    }
    const int st_in = p.status[inst];
    const int nz = p.nz, nzs = p.nzs, R = p.R, Rs = p.Rs;
#ifdef ALMPC_STAMPS
    const int stamp_id = inst + (QUEUE ? p.batch : 0);
#define PGEN_STAMP(SLOT) ALMPC_STAMP(stamp_id, SLOT)
    long long pg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pg_t = 0;   // main-loop phase accumulators (second region of the stamp buffer)
#define PG_T0() do { __builtin_amdgcn_sched_barrier(0); pg_t = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PG_ACC(S) do { __builtin_amdgcn_sched_barrier(0); const long long t_ = __builtin_readcyclecounter(); pg_acc[S] += t_ - pg_t; pg_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PGEN_STAMP(SLOT)
#define PG_T0()
#define PG_ACC(S)
#endif
    PGEN_STAMP(0);
    const double* Gh = p.Ghat + (size_t)inst * p.Ghat_stride;
    const double* gnv = p.gnorm + (size_t)inst * p.gnorm_stride;
    const double* dvi = p.dvec + (size_t)inst * p.d_stride;
    const double* Ai = p.roll.A + (size_t)inst * p.A_stride;
    const double* Bi = p.roll.B + (size_t)inst * p.B_stride;
    const int n_ = p.n, m_ = p.m, N_ = p.N;
    const double* lA = p.ltv ? p.ltvA + (size_t)inst * N_ * n_ * n_ : nullptr;
    const double* lB = p.ltv ? p.ltvB + (size_t)inst * N_ * n_ * m_ : nullptr;
    const double* lC = (p.ltv && p.ltvC) ? p.ltvC + (size_t)inst * N_ * n_ : nullptr;
    auto roll = [&](double* Zb) __attribute__((always_inline)) {
        if (p.ltv) {
            switch (p.roll_cpl) {
                case 1: rollout_steps_ltv<1>(Zb, n_, m_, N_, p.roll_g, lane, lA, lB, lC); break;
                case 2: rollout_steps_ltv<2>(Zb, n_, m_, N_, p.roll_g, lane, lA, lB, lC); break;
                case 4: rollout_steps_ltv<4>(Zb, n_, m_, N_, p.roll_g, lane, lA, lB, lC); break;
                default: rollout_steps_ltv<8>(Zb, n_, m_, N_, p.roll_g, lane, lA, lB, lC); break;
            }
        } else {
            switch (p.roll_cpl) {
                case 1: rollout_steps<1>(Zb, n_, m_, N_, p.roll_g, lane, Ai, Bi); break;
                case 2: rollout_steps<2>(Zb, n_, m_, N_, p.roll_g, lane, Ai, Bi); break;
                case 4: rollout_steps<4>(Zb, n_, m_, N_, p.roll_g, lane, Ai, Bi); break;
                default: rollout_steps<8>(Zb, n_, m_, N_, p.roll_g, lane, Ai, Bi); break;
            }
        }
    };
    const size_t base = (size_t)inst * nzs;
    const int pos = lane & (WL - 1), hf = (HS == 2) ? (lane >> 5) : 0;
    const bool lowhalf = lane < WL;

    double* Sl = smem + (size_t)wv * pgen_lds_per_wave(WL);
    double* rowbuf = Sl + WL * WL;  // [Rs]
    double* s0buf = rowbuf + 512;   // [Rs] s0 (row values of the unconstrained minimiser): read by the confirmation only, so not in registers
    double* pbufa = s0buf + 512;
    double* pbufb = pbufa + WL;
    int* wrow_s = reinterpret_cast<int*>(pbufb + WL);
    const RolloutParams& rp = p.roll;
    const int n = p.n, m = p.m, N = p.N, C = n + m;
    double* Z = Sl;  // trajectory buffer (N+1) x C, first use (s0) and last use (outputs); Sinv lives here in between
    double* EUB = Sl;         // blocked rollout of the outputs: e_u in stage order, zero padded (256 doubles)
    const bool blocked = !p.ltv && p.rollM != nullptr && p.A_stride == 0;

    // ---- rows of this lane: pairs (2 lane, 2 lane + 1) + 128 q
    int rrow[NP][2];
    double sv[NP][2], lo[NP][2], hi[NP][2], ign[NP][2];   // ign: 1 / sqrt(Ghat_rr), the violation measure's scale (pad rows: 1)
    unsigned actm = 0, eqm = 0;   // bit 2 q + e: the row is in the working set / is a terminal-equality row
    double y0 = 0, y1 = 0, z0 = 0, z1 = 0;
    bool x0_bad = false;
    // unconstrained minimiser v0 and the box-ADMM iterate (input rows live in pair 0)
    {
        double s0v[NP][2], gn[NP][2];
        const int rc = (2 * lane < nzs) ? 2 * lane : 0;
        const d2 vv = *reinterpret_cast<const d2*>(p.v0 + base + rc);
        const d2 yy = *reinterpret_cast<const d2*>(p.ys + base + rc);
        const d2 zz = *reinterpret_cast<const d2*>(p.zs + base + rc);
        const d2 dv = *reinterpret_cast<const d2*>(dvi + rc);
        y0 = yy[0]; y1 = yy[1]; z0 = zz[0]; z1 = zz[1];
        // trajectory of v0: Z rows [e_x(k); e_u(k)]
        for (int e = 0; e < 2; ++e) {
            const int r = 2 * lane + e;
            if (r < nz) Z[(size_t)(r / m) * C + n + r % m] = vv[e] * dv[e];
        }
        for (int i = lane; i < n; i += 64) {
            // (time-varying design: stage 1 of the linearisation trajectory, which sits in the reference slot, is x0; dx_0 = 0)
            const double xv = p.ltv ? rp.xref[(size_t)inst * rp.xref_stride + i] : rp.x0[(size_t)inst * n + i];
            Z[i] = p.ltv ? 0.0 : xv - rp.xref[(size_t)inst * rp.xref_stride + i];
            if (p.has_box && !(xv >= p.xmin[i] && xv <= p.xmax[i])) x0_bad = true;  // stage 1 is x0 itself
        }
        // Row data: every load is unconditional on a clamped index and issued BEFORE the rollout of v0, level by level (tables, then
        // what they point to), so that the whole set is in flight together and its latency hides behind the rollout.
        int tr_[NP][2], eq_[NP][2], xi_[NP][2], si_[NP][2];
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * lane + e + 128 * q;
                const int rcl = r < R ? r : 0;
                rrow[q][e] = r;
                tr_[q][e] = p.row_traj[rcl]; eq_[q][e] = p.row_eq[rcl]; xi_[q][e] = p.row_xidx[rcl]; si_[q][e] = p.row_state[rcl];
                gn[q][e] = gnv[rcl];
            }
        double xr_[NP][2], xmn_[NP][2], xmx_[NP][2], eqo_[NP][2], urf[2], umn[2], umx[2];
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                xr_[q][e] = rp.xref[(size_t)inst * rp.xref_stride + xi_[q][e]];
                xmn_[q][e] = p.xmin[si_[q][e]]; xmx_[q][e] = p.xmax[si_[q][e]];
                eqo_[q][e] = p.eq_off ? -p.eq_off[(size_t)inst * p.eq_stride + si_[q][e]] : 0.0;
            }
#pragma unroll
        for (int e = 0; e < 2; ++e) {   // input rows live in pair 0
            const int r = 2 * lane + e, rcl = r < nz ? r : 0;
            urf[e] = p.uref[(size_t)inst * p.uref_stride + rcl];
            umn[e] = p.umin[rcl % m]; umx[e] = p.umax[rcl % m];
        }
        wave_fence_lds();
        PGEN_STAMP(11);
        const bool use_basis = p.s0_basis != nullptr;
        double s0b[NP][2];
        if (!use_basis) roll(Z);
        else {   // s0 of the lane's rows = basis[n] + sum_j e0_j basis[j]  (e0 sits in Z[0 .. n))
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int off = 2 * lane + 128 * q;
                const d2 b = *reinterpret_cast<const d2*>(p.s0_basis + (size_t)n * Rs + (off < Rs ? off : 0));
                s0b[q][0] = b[0]; s0b[q][1] = b[1];
            }
            for (int j0 = 0; j0 < n; j0 += 4) {   // (four basis rows = 4 NP loads in flight)
                d2 bj[4][NP];
                double ej[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + u < n ? j0 + u : n - 1;
                    ej[u] = j0 + u < n ? Z[j] : 0.0;
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const int off = 2 * lane + 128 * q;
                        bj[u][q] = *reinterpret_cast<const d2*>(p.s0_basis + (size_t)j * Rs + (off < Rs ? off : 0));
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < NP; ++q) { s0b[q][0] = fma(ej[u], bj[u][q][0], s0b[q][0]); s0b[q][1] = fma(ej[u], bj[u][q][1], s0b[q][1]); }
            }
        }
        PGEN_STAMP(12);
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = rrow[q][e];
                const bool valid = r < R;
                if (!valid) gn[q][e] = 1.0;
                if (valid && r >= nz && eq_[q][e] != 0) eqm |= 1u << (2 * q + e);
                const double zst = use_basis ? s0b[q][e] : Z[tr_[q][e]];   // state row: value e_x[i,k] of the v0 trajectory (row 0 of Z for the others)
                if (q == 0 && r < nz) {  // input row, bounds exactly as k_admm forms them
                    const double di = 1.0 / (e ? dv[1] : dv[0]);
                    lo[q][e] = (umn[e] - urf[e]) * di;
                    hi[q][e] = (umx[e] - urf[e]) * di;
                    s0v[q][e] = e ? vv[1] : vv[0];
                } else if (valid) {  // state row, shared bounds relative to the reference (or the linearisation trajectory)
                    s0v[q][e] = zst;
                    lo[q][e] = eq_[q][e] ? eqo_[q][e] : xmn_[q][e] - xr_[q][e];
                    hi[q][e] = eq_[q][e] ? eqo_[q][e] : xmx_[q][e] - xr_[q][e];
                } else { lo[q][e] = -__builtin_inf(); hi[q][e] = __builtin_inf(); s0v[q][e] = 0.0; }
                sv[q][e] = s0v[q][e];
            }
        if (p.eq_proj) {   // start from the minimiser subject to the terminal equality; its rows leave the problem
            const int ne = p.ne, eq0 = p.eq0;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                d2 v; v[0] = s0v[q][0]; v[1] = s0v[q][1];
                *reinterpret_cast<d2*>(rowbuf + 2 * lane + 128 * q) = v;
            }
            wave_fence_lds();
            double lamE = 0.0;
            {   // (same sum, four terms' loads in flight per trip; lanes >= ne read row 0 and drop the result)
                const int lr = lane < ne ? lane : 0;
                for (int e0_ = 0; e0_ < ne; e0_ += 4) {
                    double wv_[4], bv_[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int e2 = e0_ + u < ne ? e0_ + u : ne - 1;
                        wv_[u] = p.WinvE[lr * ne + e2];
                        bv_[u] = p.eq_off ? -p.eq_off[(size_t)inst * p.eq_stride + e2] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (e0_ + u < ne) lamE += wv_[u] * (rowbuf[eq0 + e0_ + u] - bv_[u]);
                }
                if (lane >= ne) lamE = 0.0;
            }
            if (lane < WL) pbufa[lane] = (lane < ne) ? lamE : 0.0;   // (ne <= n <= 32 <= WL)
            wave_fence_lds();
            for (int e0_ = 0; e0_ < ne; e0_ += 4) {   // (four rows of GhatE = 4 NP loads in flight: one row per trip was an L2 round trip per row)
                d2 g[4][NP];
                double le[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e2 = e0_ + u < ne ? e0_ + u : ne - 1;
                    le[u] = e0_ + u < ne ? pbufa[e2] : 0.0;
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const int off = 2 * lane + 128 * q;
                        g[u][q] = *reinterpret_cast<const d2*>(p.GhatE + (size_t)e2 * Rs + (off < Rs ? off : 0));
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < NP; ++q) { s0v[q][0] -= g[u][q][0] * le[u]; s0v[q][1] -= g[u][q][1] * le[u]; }
            }
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int r = rrow[q][e];
                    if (r >= eq0 && r < eq0 + ne) { s0v[q][e] = lo[q][e]; lo[q][e] = -__builtin_inf(); hi[q][e] = __builtin_inf(); }
                    sv[q][e] = s0v[q][e];
                }
            eqm = 0;
            wave_fence_lds();
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            d2 v; v[0] = s0v[q][0]; v[1] = s0v[q][1];
            *reinterpret_cast<d2*>(s0buf + 2 * lane + 128 * q) = v;
            ign[q][0] = 1.0 / gn[q][0]; ign[q][1] = 1.0 / gn[q][1];
        }
    }
    x0_bad = __any(x0_bad);
    wave_fence_lds();
    PGEN_STAMP(1);
    // ---- Sinv := identity (Z is dead now)
#pragma unroll 8
    for (int t = 0; t < WL / HS; ++t) Sl[(HS * t + hf) * WL + pos] = ((HS * t + hf) == pos) ? 1.0 : 0.0;
    wrow_s[pos] = 0;
    wave_fence_lds();

    int wrow = 0, wsd = 0;
    double wbnd = 0.0, lam = 0.0;
    int k = 0;
    int fin = 1;  // 0 optimal, 1 cap / capacity, 3 infeasible
    int it = 0;

    auto put_pos = [&](double* buf, double v) __attribute__((always_inline)) { buf[pos] = v; wave_fence_lds(); };
    auto put_rows = [&](const double (&a)[NP][2]) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            d2 v; v[0] = a[q][0]; v[1] = a[q][1];
            *reinterpret_cast<d2*>(rowbuf + 2 * lane + 128 * q) = v;
        }
        wave_fence_lds();
    };
    auto s_matvec = [&](const double* cb) __attribute__((always_inline)) -> double {
        double acc = 0.0;
        for (int l0 = 0; l0 < k; l0 += 8 * HS) {
            double a[8], c[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) { a[t] = Sl[(l0 + HS * t + hf) * WL + pos]; c[t] = cb[l0 + HS * t + hf]; }
#pragma unroll
            for (int t = 0; t < 8; ++t) acc += a[t] * c[t];
        }
        if (HS == 2) acc += __shfl_xor(acc, 32);
        return acc;
    };
    auto s_rank1 = [&](double a, const double* ab, double scale) __attribute__((always_inline)) {
        const double as = a * scale;
        for (int l0 = 0; l0 < k; l0 += 8 * HS) {
            double cur[8], av[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) { cur[t] = Sl[(l0 + HS * t + hf) * WL + pos]; av[t] = ab[l0 + HS * t + hf]; }
#pragma unroll
            for (int t = 0; t < 8; ++t) Sl[(l0 + HS * t + hf) * WL + pos] = cur[t] + as * av[t];
        }
    };
    // Rows of Ghat for the working set, CH at a time.  (Measured and dropped: a second group in flight, and group 0 requested as soon
    // as W is final so that it travels under the next iteration's scan -- 64 more live registers each, and the NP = 4 build, at 256
    // already, answers with 0.5 - 1.2 KB of scratch per lane; in the second launch's build, which may use 512, the compiler parks the
    // arriving group in AGPRs behind a full wait and the iteration gets slower, 14 k -> 16 k cycles.)
    // (a in LDS buffer ab, zero beyond k)
    auto g_rows_minus = [&](const double* ab, double (&qv)[NP][2]) {   // qv[.] -= sum_{l<k} Ghat[W_l, rows] * a_l
        if constexpr (COOP) {
            // rows l = 0, 4, 8, ... here; l = w, w + 4, ... on helper wave w, whose partial sums come back through LDS.  Two workgroup
            // barriers per call; the coefficients are in pbufb (the only buffer this is ever called with), the rows in wrow_s.
            int* job = reinterpret_cast<int*>(smem + pgen_lds_per_wave(WL));
            double* part = smem + pgen_lds_per_wave(WL) + 2;
            if (lane == 0) { job[0] = k; job[1] = inst; }
            __syncthreads();
            constexpr int NW = PGEN_HELPERS + 1;
            for (int l0 = 0; l0 < k; l0 += NW * CH) {
                d2 g[CH][NP];
                double av[CH];
#pragma unroll
                for (int t = 0; t < CH; ++t) {
                    const int l = l0 + NW * t;
                    const int lc = l < k ? l : 0;
                    const double* row = Gh + (size_t)__builtin_amdgcn_readlane(wrow, lc) * Rs;
                    av[t] = l < k ? ab[lc] : 0.0;
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const int off = 2 * lane + 128 * q;
                        g[t][q] = *reinterpret_cast<const d2*>(row + (off < Rs ? off : 0));
                    }
                }
#pragma unroll
                for (int t = 0; t < CH; ++t)
#pragma unroll
                    for (int q = 0; q < NP; ++q) { qv[q][0] -= g[t][q][0] * av[t]; qv[q][1] -= g[t][q][1] * av[t]; }
            }
            __syncthreads();
#pragma unroll
            for (int w = 0; w < PGEN_HELPERS; ++w)
#pragma unroll
                for (int q = 0; q < NP; ++q) {   // (helper 1 writes into rowbuf -- free while this runs --, 2 and 3 behind the job word)
                    const double* src = w == 0 ? rowbuf : part + (size_t)(w - 1) * 512;
                    const d2 t = *reinterpret_cast<const d2*>(src + 2 * lane + 128 * q);
                    qv[q][0] -= t[0]; qv[q][1] -= t[1];
                }
            return;
        }
        for (int l0 = 0; l0 < k; l0 += CH) {   // (all NP * CH loads of a group go out before the first is used)
            d2 g[CH][NP];
            double av[CH];
#pragma unroll
            for (int t = 0; t < CH; ++t) {
                // (row index from the position's register: a scalar base address, one lane offset per pair for all rows)
                const double* row = Gh + (size_t)__builtin_amdgcn_readlane(wrow, l0 + t) * Rs;
                av[t] = ab[l0 + t];
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const int off = 2 * lane + 128 * q;
                    g[t][q] = *reinterpret_cast<const d2*>(row + (off < Rs ? off : 0));
                }
            }
#pragma unroll
            for (int t = 0; t < CH; ++t)
#pragma unroll
                for (int q = 0; q < NP; ++q) { qv[q][0] -= g[t][q][0] * av[t]; qv[q][1] -= g[t][q][1] * av[t]; }
        }
    };
    auto load_row = [&](int j, double (&gj)[NP][2]) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int off = 2 * lane + 128 * q;
            const d2 g = *reinterpret_cast<const d2*>(Gh + (size_t)j * Rs + (off < Rs ? off : 0));
            gj[q][0] = g[0]; gj[q][1] = g[1];
        }
    };
    auto row_value = [&](const double (&a)[NP][2], int j) -> double {  // a[row j], uniform j
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            // (both elements read, then a scalar select: `(j & 1) ? a[q][1] : a[q][0]` in front of the readlane made the compiler keep a
            // copy of sv / lo / hi in scratch memory and index it -- three scratch loads per selection, a third of the iteration)
            const double c0 = readlane_d(a[q][0], (j >> 1) & 63), c1 = readlane_d(a[q][1], (j >> 1) & 63);
            if ((j >> 7) == q) v = (j & 1) ? c1 : c0;
        }
        return v;
    };
    auto recompute = [&]() __attribute__((always_inline)) {  // lam = Sinv (s0_W - b), s = s0 - Ghat[:,W] lam
        const double rv = s0buf[wrow] - wbnd;
        put_pos(pbufa, (pos < k) ? rv : 0.0);
        const double lm = s_matvec(pbufa);
        lam = (pos < k) ? lm : 0.0;
        put_pos(pbufb, lam);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const d2 v = *reinterpret_cast<const d2*>(s0buf + 2 * lane + 128 * q);
            sv[q][0] = v[0]; sv[q][1] = v[1];
        }
        g_rows_minus(pbufb, sv);
    };
    // The rows of the working set must sit on their bounds in values computed from scratch.  With an ill-conditioned Ghat_WW the
    // bordered inverse loses digits and they do not: refine the multipliers on that residual (a few rounds of iterative refinement);
    // false when it does not settle -- the instance is then NOT reported as solved.
    auto refine = [&]() -> bool {
        put_rows(ign);
        const double igw = rowbuf[wrow];
        wave_fence_lds();
        for (int rr = 0; rr < 4; ++rr) {
            put_rows(sv);
            const double rv = rowbuf[wrow] - wbnd;
            const double res = wave_max((lowhalf && pos < k) ? fabs(rv) * igw : 0.0);
            if (res <= 1e-9) return true;
            if (rr == 3) break;
            put_pos(pbufa, (pos < k) ? rv : 0.0);
            const double dl = s_matvec(pbufa);
            const double dlm = (pos < k) ? dl : 0.0;
            lam += dlm;
            put_pos(pbufb, dlm);
            g_rows_minus(pbufb, sv);
        }
        return false;
    };
    auto mark = [&](int j, bool on, double bval) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (rrow[q][e] == j) actm = on ? (actm | (1u << (2 * q + e))) : (actm & ~(1u << (2 * q + e)));
    };
    // border Sinv with row j given u = Sinv c and isc = 1/(Ghat_jj - c'u)
    auto border = [&](int j, double u, double isc, double bval, int sd, double lamj) __attribute__((always_inline)) {
        s_rank1(u, pbufb, isc);  // pbufb holds u
        const double bv = (pos == k) ? isc : -u * isc;
        Sl[k * WL + pos] = bv;
        Sl[pos * WL + k] = bv;
        if (pos == k) { wrow = j; wsd = sd; wbnd = bval; lam = lamj; }
        if (lane == 0) wrow_s[k] = j;
        mark(j, true, bval);
        k = __builtin_amdgcn_readfirstlane(k + 1);
        wave_fence_lds();
    };
    // remove position rp; with_state: move s and lam so that lam_rp becomes 0 first (used to restore dual feasibility)
    auto remove_pos = [&](int rp, bool with_state) __attribute__((always_inline)) {
        const double sp = Sl[rp * WL + pos];
        const double spp = readlane_d(sp, rp);
        put_pos(pbufb, sp);
        if (with_state) {
            const double a = readlane_d(lam, rp) / spp;
            double qv[NP][2];
#pragma unroll
            for (int q = 0; q < NP; ++q) { qv[q][0] = 0.0; qv[q][1] = 0.0; }
            g_rows_minus(pbufb, qv);
#pragma unroll
            for (int q = 0; q < NP; ++q) { sv[q][0] -= a * qv[q][0]; sv[q][1] -= a * qv[q][1]; }
            lam -= sp * a;
        }
        s_rank1(sp, pbufb, -1.0 / spp);
        wave_fence_lds();
        const int last = k - 1;
        const int jrem = __builtin_amdgcn_readlane(wrow, rp);
        if (rp != last) {
            const double colv = Sl[last * WL + pos];
            const double corner = readlane_d(colv, last);
            const double nv = (pos == rp) ? corner : ((pos == last) ? 0.0 : colv);
            Sl[rp * WL + pos] = nv;
            Sl[pos * WL + rp] = nv;
            const int lrow = __builtin_amdgcn_readlane(wrow, last), lsd = __builtin_amdgcn_readlane(wsd, last);
            const double lbv = readlane_d(wbnd, last), llam = readlane_d(lam, last);
            if (pos == rp) { wrow = lrow; wsd = lsd; wbnd = lbv; lam = llam; }
            if (lane == 0) wrow_s[rp] = lrow;
        }
        const double iv = (pos == last) ? 1.0 : 0.0;
        Sl[last * WL + pos] = iv;
        Sl[pos * WL + last] = iv;
        if (pos == last) lam = 0.0;
        if (lane == 0) wrow_s[last] = 0;
        mark(jrem, false, 0.0);
        k = __builtin_amdgcn_readfirstlane(k - 1);
        wave_fence_lds();
    };
    // direction of adding row j, first half: c = Ghat[W,j] -> u = Sinv c (also left in pbufb), dp = Ghat_jj - c'u
    auto dir_u = [&](int j, double (&gj)[NP][2], double& u, double& dp, double& gjj, bool loaded = false) __attribute__((always_inline)) {
        double cv, gd;
        if (!loaded) {
            // main loop: c = Ghat[j, W] (symmetry) gathered straight from the row in L2 next to the row's own loads -- no trip through LDS
            load_row(j, gj);
            cv = Gh[(size_t)j * Rs + wrow];
            gd = Gh[(size_t)j * Rs + j];
        } else {   // guess build: the row was requested one step ahead, the working set has changed since
            put_rows(gj);
            cv = rowbuf[wrow];
            gd = rowbuf[j];
        }
        const double c = (pos < k) ? cv : 0.0;
        gjj = readlane_d(gd, 0);
        put_pos(pbufa, c);
        u = s_matvec(pbufa);
        put_pos(pbufb, u);
        dp = gjj - wave_sum(lowhalf ? c * u : 0.0);
    };
    // second half: d = Ghat[:,j] - Ghat[:,W] u  (u in pbufb)
    auto dir_d = [&](double (&gj)[NP][2]) __attribute__((always_inline)) { g_rows_minus(pbufb, gj); };   // in place: gj becomes d

    const bool skip = (st_in == 2) || x0_bad;
    bool give_up = skip;
    bool overflow = false;  // the working set hit WL rows
    bool guess_overflow = false;  // ... while the guess was still being built (its rows beyond WL are not in the working set)
    if (!skip) {
        // ---- initial working set: equality rows, then the input rows the ADMM multipliers flag (rows ascending)
        int flag[NP][2];
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int f = 0;
                const int r = rrow[q][e];
                if (r < R) {
                    if (r >= nz) { if ((eqm >> (2 * q + e)) & 1u) f = 2; }
                    else {
                        const double yy = e ? y1 : y0, ww = fmin(fmax(e ? z1 : z0, lo[q][e]), hi[q][e]);
                        if (yy < 0.0 && ww <= lo[q][e]) f = -1;
                        else if (yy > 0.0 && ww >= hi[q][e]) f = 1;
                    }
                }
                flag[q][e] = f;
            }
        // positions: equality rows first (in row order), then flagged input rows (row order)
        int* ibuf = reinterpret_cast<int*>(rowbuf);  // [0..64) row of position, [64..128) side
        int base_pos = 0;
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const bool f0 = pass == 0 ? flag[q][0] == 2 : (flag[q][0] == 1 || flag[q][0] == -1);
                const bool f1 = pass == 0 ? flag[q][1] == 2 : (flag[q][1] == 1 || flag[q][1] == -1);
                const unsigned long long m0 = __ballot(f0), m1 = __ballot(f1);
                const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
                const int p0 = base_pos + __popcll(m0 & below) + __popcll(m1 & below), p1 = p0 + (f0 ? 1 : 0);
                if (f0 && p0 < 64) { ibuf[p0] = rrow[q][0]; ibuf[64 + p0] = pass == 0 ? 0 : flag[q][0]; }
                if (f1 && p1 < 64) { ibuf[p1] = rrow[q][1]; ibuf[64 + p1] = pass == 0 ? 0 : flag[q][1]; }
                base_pos += __popcll(m0) + __popcll(m1);
            }
        }
        cnt = base_pos;
        wave_fence_lds();
        // guessed rows live in registers of lane = guess index (up to 64 of them)
        int g_row = (lane < cnt && lane < 64) ? ibuf[lane] : 0;
        int g_sd = (lane < cnt && lane < 64) ? ibuf[64 + lane] : 0;
        if constexpr (QUEUE) {   // second launch: start from the working set the first launch had when it ran out of room
            const int w0 = p.ovf_ws[(size_t)inst * 32];
            if (w0 >= 0) {
                const int wv_ = p.ovf_ws[(size_t)inst * 32 + (lane & 31)];
                cnt = 32;
                g_row = lane < 32 ? (wv_ & 0xFFF) : 0;
                g_sd = lane < 32 ? ((wv_ >> 12) & 3) - 1 : 0;
                if (p.ovf_sinv) {   // ... and from its inverse: nothing to border in again
                    const double* sv_ = p.ovf_sinv + (size_t)inst * (32 * 32 + 32);
                    double col[32];
#pragma unroll
                    for (int t = 0; t < 32; ++t) col[t] = sv_[t * 32 + (lane & 31)];
                    const double bnd_ = sv_[32 * 32 + (lane & 31)];
                    if (lane < 32) {
#pragma unroll
                        for (int t = 0; t < 32; ++t) Sl[t * WL + lane] = col[t];
                        wrow = g_row; wsd = g_sd; wbnd = bnd_;
                        wrow_s[lane] = g_row;
                    }
                    for (int i = 0; i < 32; ++i) mark(__builtin_amdgcn_readlane(g_row, i), true, 0.0);
                    k = 32;
                    cnt = 0;
                    wave_fence_lds();
                }
            }
        }
        wave_fence_lds();
        // Build the working set row by row (bordering), skipping rows that are linearly dependent on the rows already
        // taken: a saturated guess can hold more rows than there are variables.  The next row of Ghat is requested while the
        // current one is processed.
        const int ng = cnt < 64 ? cnt : 64;
        double gnx[NP][2];
        if (ng > 0) load_row(__builtin_amdgcn_readlane(g_row, 0), gnx);
        for (int gi = 0; gi < ng; ++gi) {
            if (k == WL) { give_up = true; overflow = true; guess_overflow = true; break; }
            const int j = __builtin_amdgcn_readlane(g_row, gi), sd = __builtin_amdgcn_readlane(g_sd, gi);
            double gj[NP][2], u, dp, gjj;
#pragma unroll
            for (int q = 0; q < NP; ++q) { gj[q][0] = gnx[q][0]; gj[q][1] = gnx[q][1]; }
            if (gi + 1 < ng) load_row(__builtin_amdgcn_readlane(g_row, gi + 1), gnx);
            dir_u(j, gj, u, dp, gjj, true);
            if (!(dp > 1e-10 * gjj)) continue;
            border(j, u, 1.0 / dp, sd < 0 ? row_value(lo, j) : row_value(hi, j), sd, 0.0);   // (equality rows: lo == hi)
        }
        PGEN_STAMP(2);
#ifdef ALMPC_STAMPS
        if (g_stamps && lane == 0) g_stamps[(size_t)stamp_id * 16 + 8] = k;
#endif
        if (!give_up && k > 0) recompute();
    }
    PGEN_STAMP(3);

    const int max_iter = p.max_iter;
    if (!give_up) {
        // ---- restore dual feasibility of the guess: drop wrong-sign multipliers, worst first
        while (k > 0 && it < max_iter) {
            const bool mine = lowhalf && pos < k && wsd != 0;
            const double viol = mine ? ((wsd > 0) ? -lam : lam) : -__builtin_inf();
            const double lmax = wave_max((lowhalf && pos < k) ? fabs(lam) : 0.0);
            const double vmax = wave_max(viol);
            if (!(vmax > 1e-12 * fmax(1.0, lmax))) break;
            const int vi = __builtin_ctzll(__ballot(mine && viol == vmax));
            remove_pos(vi, true);
            ++it;
        }
        PGEN_STAMP(4);
        // ---- Goldfarb-Idnani main loop
        bool fresh = false;
        while (it < max_iter) {
            ++it;
            PG_T0();
            // most violated row outside W, in the H'^-1 metric
            // (each lane keeps side and bound of its own best row: the owner's are read with three v_readlane pairs, where picking them
            // out of sv / hi / lo by the row number took 48)
            double vbest = -__builtin_inf(), bbest = 0.0;
            int which = 0;
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    // (rows beyond R have infinite bounds: their measure is -inf by itself)
                    const double over = sv[q][e] - hi[q][e], under = lo[q][e] - sv[q][e];
                    const double v = !((actm >> (2 * q + e)) & 1u) ? fmax(over, under) * ign[q][e] : -__builtin_inf();
                    if (v > vbest) { vbest = v; which = (2 * q + e) | (over > 0.0 ? 16 : 0); bbest = over > 0.0 ? hi[q][e] : lo[q][e]; }
                }
            const double vmax = wave_max(vbest);
            if (!(vmax == vmax) || !(vmax < __builtin_inf())) { fin = 1; break; }  // numerical breakdown: never report it as solved
            if (!(vmax > 1e-9)) {
                if (fresh) { fin = 0; break; }
                recompute();  // confirm on values computed from scratch
                const bool ref_ok = !(k > 0) || refine();
                PG_ACC(6);
                if (!ref_ok) { fin = 1; break; }   // (measured: rebuilding the inverse from the rows does not rescue these --
                                                              // their Ghat_WW is singular to working precision: edge of feasibility)
                fresh = true;
                --it;
                continue;
            }
            fresh = false;
            const int owner = __builtin_ctzll(__ballot(vbest == vmax));
            const int wsel = __builtin_amdgcn_readlane(which, owner);
            const int pr = 2 * owner + (wsel & 1) + 128 * ((wsel & 15) >> 1);
            const int sd = (wsel & 16) ? 1 : -1;
            const double bp = readlane_d(bbest, owner);
            double lam_p = 0.0;
            bool added = false;
            PG_ACC(0);
            while (it < max_iter) {
                if (k == WL) { fin = 1; give_up = true; overflow = true; break; }
                double u, dp, gjj, dv[NP][2];
                dir_u(pr, dv, u, dp, gjj);
                PG_ACC(1);
                dir_d(dv);
                PG_ACC(2);
                const bool dependent = !(dp > 1e-12 * gjj);
                const double spn = row_value(sv, pr);
                const double tau2 = dependent ? (double)sd * __builtin_inf() : (spn - bp) / dp;
                // multipliers that would change sign on the way: lam_i - u_i tau = 0
                const bool cand = lowhalf && pos < k && wsd != 0 && u != 0.0;
                const double ti = lam / u;
                const bool zero_blk = cand && lam == 0.0 && (u * sd * wsd > 0.0);
                const bool pos_blk = cand && lam != 0.0 && (ti * sd > 0.0);
                const double tabs = zero_blk ? 0.0 : (pos_blk ? fabs(ti) : __builtin_inf());
                const double tau1 = wave_min(tabs);
                PG_ACC(3);
                if (dependent && !(tau1 < __builtin_inf())) { fin = 3; break; }
                if (tau1 < fabs(tau2)) {  // partial step, drop the blocking row, try again
                    const double tau = sd * tau1;
                    if (!dependent) {
#pragma unroll
                        for (int q = 0; q < NP; ++q) { sv[q][0] -= tau * dv[q][0]; sv[q][1] -= tau * dv[q][1]; }
                    }
                    lam -= u * tau;
                    lam_p += tau;
                    const int blk = __builtin_ctzll(__ballot(tabs == tau1));
                    if (pos == blk) lam = 0.0;
                    remove_pos(blk, false);
                    ++it;
                    PG_ACC(4);
                    continue;
                }
#pragma unroll
                for (int q = 0; q < NP; ++q) { sv[q][0] -= tau2 * dv[q][0]; sv[q][1] -= tau2 * dv[q][1]; }
                lam -= u * tau2;
                lam_p += tau2;
                border(pr, u, 1.0 / dp, bp, sd, lam_p);
                added = true;
                PG_ACC(5);
                break;
            }
            if (!added) break;
        }
    }
    if (x0_bad && st_in != 2) fin = 3;
    PGEN_STAMP(5);
#ifdef ALMPC_STAMPS
    if (g_stamps && lane == 0) {
        g_stamps[(size_t)stamp_id * 16 + 9] = it; g_stamps[(size_t)stamp_id * 16 + 10] = k;
        for (int i_ = 0; i_ < 8; ++i_) g_stamps[(size_t)(2 * p.batch + stamp_id) * 16 + i_] = pg_acc[i_];
    }
#endif

    // ---- result: w = input rows of s (pair 0), then the fused rollout
    d2 wout;
    if (give_up || fin == 3) {
        wout[0] = (st_in == 2) ? z0 : fmin(fmax(z0, lo[0][0]), hi[0][0]);
        wout[1] = (st_in == 2) ? z1 : fmin(fmax(z1, lo[0][1]), hi[0][1]);
    } else {
        // an input row of the working set sits on the bound it was added at: the nearer one, exactly
        const double c0 = fmin(fmax(sv[0][0], lo[0][0]), hi[0][0]), c1 = fmin(fmax(sv[0][1], lo[0][1]), hi[0][1]);
        wout[0] = (actm & 1u) ? ((sv[0][0] - lo[0][0] < hi[0][0] - sv[0][0]) ? lo[0][0] : hi[0][0]) : c0;
        wout[1] = (actm & 2u) ? ((sv[0][1] - lo[0][1] < hi[0][1] - sv[0][1]) ? lo[0][1] : hi[0][1]) : c1;
    }
    if (lane == 0) {
        p.piters[inst] = it;
        const int st_out = (st_in == 2) ? 2 : ((fin == 0) ? 0 : (fin == 3 ? 3 : 1));
        p.status[inst] = st_out;
        const bool requeue = !QUEUE && overflow && fin != 3;   // (the second launch decides about this one)
        if (requeue) p.ovf[2 + atomicAdd(p.ovf, 1)] = inst;
        if (p.unsolved && st_out == 1 && !requeue) {
            __hip_atomic_fetch_add(p.unsolved, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (p.redo_gate) *p.redo_gate = p.step_serial;
        }
    }
    if (p.redo_ws) {
        const int st_out = (st_in == 2) ? 2 : ((fin == 0) ? 0 : (fin == 3 ? 3 : 1));
        const bool requeue = !QUEUE && overflow && fin != 3;
        int32_t* rw = p.redo_ws + (size_t)inst * 64;
        // (also when the finish GAVE UP for want of room: its rows -- the full capacity of them -- are a valid working set, and the
        // instance that needs 65 rows is exactly the one whose redo from scratch costs a millisecond)
        if (st_out == 1 && !requeue && k > 0) {
            const int pos = WL == 32 ? (lane & 31) : lane;
            if (lane < WL && pos < k && pos < 63) {
                const int xi = p.row_xidx[wrow];
                const int t = wrow < p.nz ? (wrow / p.m) * p.redo_sp + p.redo_nt + wrow % p.m : (xi / p.n) * p.redo_sp + xi % p.n;
                rw[1 + pos] = (t << 2) | (wsd == 0 ? 2 : (wsd > 0 ? 1 : 0));
            }
            if (lane == 0) rw[0] = k < 63 ? k : 63;
        } else if (lane == 0) rw[0] = 0;
    }
    if (!QUEUE && overflow && fin != 3) {   // (WL = 32 here: both half-waves mirror the positions)
        if (lane < 32) p.ovf_ws[(size_t)inst * 32 + lane] = guess_overflow ? -1 : (wrow | ((wsd + 1) << 12));
        if (p.ovf_sinv && !guess_overflow) {
            double* sv_ = p.ovf_sinv + (size_t)inst * (32 * 32 + 32);
#pragma unroll 4
            for (int t = 0; t < 32; t += 2) sv_[(t + (lane >> 5)) * 32 + (lane & 31)] = Sl[(t + (lane >> 5)) * WL + (lane & 31)];
            if (lane < 32) sv_[32 * 32 + lane] = wbnd;
        }
    }
    wave_fence_lds();
    {
        const int r0 = 2 * lane, r1 = 2 * lane + 1;
        const int rc = (r0 < nzs) ? r0 : 0;
        const d2 dvp = *reinterpret_cast<const d2*>(dvi + rc);
        if (r0 < nz) {
            const double ur = rp.uref[(size_t)inst * rp.uref_stride + r0];
            const double uu = fmin(fmax(wout[0] * dvp[0] + ur, rp.umin[r0 % m]), rp.umax[r0 % m]);
            rp.u[(size_t)inst * nz + r0] = uu;
            rp.eu[(size_t)inst * nz + r0] = uu - ur;
            Z[(size_t)(r0 / m) * C + n + r0 % m] = uu - ur;
        }
        if (r1 < nz) {
            const double ur = rp.uref[(size_t)inst * rp.uref_stride + r1];
            const double uu = fmin(fmax(wout[1] * dvp[1] + ur, rp.umin[r1 % m]), rp.umax[r1 % m]);
            rp.u[(size_t)inst * nz + r1] = uu;
            rp.eu[(size_t)inst * nz + r1] = uu - ur;
            Z[(size_t)(r1 / m) * C + n + r1 % m] = uu - ur;
        }
        if (blocked) {   // outputs straight from the blocked rollout (values leave for HBM as they are produced)
            {
                d2 ev;
                ev[0] = (r0 < nz) ? Z[(size_t)(r0 / m) * C + n + r0 % m] : 0.0;
                ev[1] = (r1 < nz) ? Z[(size_t)(r1 / m) * C + n + r1 % m] : 0.0;
                wave_fence_lds();
                *reinterpret_cast<d2*>(EUB + 2 * lane) = ev;
                d2 zz; zz[0] = 0.0; zz[1] = 0.0;
                *reinterpret_cast<d2*>(EUB + 128 + 2 * lane) = zz;
            }
            double e0v = 0.0, x0r = 0.0;
            if (lane < n) { x0r = rp.x0[(size_t)inst * n + lane]; e0v = x0r - rp.xref[(size_t)inst * rp.xref_stride + lane]; }
            double cu[ROLL_SMX], cx[ROLL_NX];
            roll_load<ROLL_SMX, ROLL_NX>(p.rollM, lane, cu, cx);
            wave_fence_lds();
            PGEN_STAMP(13);
            const size_t xo = (size_t)inst * n * (N + 1);
            roll_run<ROLL_SMX, ROLL_NX>(cu, cx, n, m, N, p.roll_s, p.roll_nb, lane, EUB, e0v, x0r, nullptr,
                                        rp.xref + (size_t)inst * rp.xref_stride, rp.x + xo, rp.ex + xo);
            PGEN_STAMP(14);
        } else if (!p.ltv) {   // (x / e_x are not defined for a time-varying design: the caller rolls the nonlinear model out)
        for (int i = lane; i < n; i += 64) Z[i] = rp.x0[(size_t)inst * n + i] - rp.xref[(size_t)inst * rp.xref_stride + i];
        wave_fence_lds();
        PGEN_STAMP(13);
        roll(Z);
        PGEN_STAMP(14);
        const int nx = n * (N + 1);
        const size_t xo = (size_t)inst * nx;
        for (int t0 = 0; t0 < nx; t0 += 256) {   // (four reference loads in flight per lane)
            double xr4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = t0 + 64 * j + lane, tc = t < nx ? t : 0;
                xr4[j] = (tc < n) ? rp.x0[(size_t)inst * n + tc] : rp.xref[(size_t)inst * rp.xref_stride + tc];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = t0 + 64 * j + lane;
                if (t < nx) {
                    const double ev = Z[(size_t)(t / n) * C + t % n];
                    rp.ex[xo + t] = ev;
                    rp.x[xo + t] = (t < n) ? xr4[j] : ev + xr4[j];
                }
            }
        }
        }
    }
    PGEN_STAMP(6);
    if (QUEUE) wave_fence_lds();  // the next instance reuses this wave's LDS slot
  } while (QUEUE);
}

// ------------------------------------------------------------------------------------------------------------------------------
// k_ghat_inst: Ghat_i = A_i G_i A_i' of ONE instance per workgroup, for designs with a model per instance (almpc_design_batched,
// the re-linearisation pipeline) or per instance and stage (almpc_design_ltv, the SQP loop).  The state rows of A_i = [I; C'_i] are
// rows of the prediction matrix times the scaling, C'_i = Gamma_i[rows] D_i; applying Gamma_i to a vector is a rollout from a zero
// state, so no prediction matrix is formed:
//   pass 0, thread = column a < nz:   roll out the input sequence D_i G_i[:, a]     -> (C' G)[r, a]  (and its transpose block)
//   pass 1, thread = state row w < mc: roll out the input sequence D_i (C' G)[w, :]' -> (C' G C)[r, w]
// with the state in registers (NX >= n of them) and the stage's [A B] broadcast from LDS.  Cold path: design time only.
struct GhatInstParams {
    int n, m, N, nz, nzs, mc, R, Rs;
    const double* A; const double* B;          // column-major blocks
    long A_stride, B_stride;                   // doubles between instances
    long A_kstride, B_kstride;                 // doubles between stages (0: one model for the whole horizon)
    const double* G; long G_stride;            // [nz][nzs] H'_i^-1
    const double* dvec; long d_stride;
    const int* rowmap;                         // [N*n]: state (stage k+2, i) at k*n + i -> state-row index 0..mc-1, or -1
    double* Ghat; long Ghat_stride;            // [R][Rs] (columns R..Rs stay zero)
    double* gnorm; long gnorm_stride;          // [Rs]
};
constexpr int GHAT_THREADS = 256;

template <int NX>
__global__ __launch_bounds__(GHAT_THREADS) void k_ghat_inst(GhatInstParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int inst = blockIdx.x, tid = threadIdx.x;
    const int n = p.n, m = p.m, N = p.N, nz = p.nz, nzs = p.nzs, mc = p.mc, R = p.R, Rs = p.Rs;
    double* AB = smem;  // A (n x n) then B (n x m), column-major
    const double* G = p.G + (size_t)inst * p.G_stride;
    const double* dv = p.dvec + (size_t)inst * p.d_stride;
    double* Gh = p.Ghat + (size_t)inst * p.Ghat_stride;
    double* gn = p.gnorm + (size_t)inst * p.gnorm_stride;
    for (int t = tid; t < nz * nz; t += GHAT_THREADS) Gh[(size_t)(t / nz) * Rs + t % nz] = G[(size_t)(t / nz) * nzs + t % nz];
    for (int pass = 0; pass < 2; ++pass) {
        const int ncol = pass == 0 ? nz : mc;
        for (int c0 = 0; c0 < ncol; c0 += GHAT_THREADS) {
            const int c = c0 + tid;
            const bool on = c < ncol;
            const int cc = on ? c : 0;
            double x[NX];
#pragma unroll
            for (int i = 0; i < NX; ++i) x[i] = 0.0;
            for (int t = 0; t < N; ++t) {
                if (t == 0 || p.A_kstride != 0) {
                    __syncthreads();
                    const double* At = p.A + (size_t)inst * p.A_stride + (size_t)t * p.A_kstride;
                    const double* Bt = p.B + (size_t)inst * p.B_stride + (size_t)t * p.B_kstride;
                    for (int e = tid; e < n * n; e += GHAT_THREADS) AB[e] = At[e];
                    for (int e = tid; e < n * m; e += GHAT_THREADS) AB[n * n + e] = Bt[e];
                    __syncthreads();
                }
                double xn[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) xn[i] = 0.0;
#pragma unroll
                for (int j = 0; j < NX; ++j)
                    if (j < n) {
                        const double xj = x[j];
#pragma unroll
                        for (int i = 0; i < NX; ++i)
                            if (i < n) xn[i] += AB[j * n + i] * xj;
                    }
                for (int b = 0; b < m; ++b) {
                    const int row = t * m + b;
                    const double src = pass == 0 ? G[(size_t)row * nzs + cc] : Gh[(size_t)(nz + cc) * Rs + row];
                    const double ub = dv[row] * src;
#pragma unroll
                    for (int i = 0; i < NX; ++i)
                        if (i < n) xn[i] += AB[n * n + b * n + i] * ub;
                }
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    x[i] = xn[i];
                    if (i < n && on) {
                        const int r = p.rowmap[t * n + i];
                        if (r >= 0) {
                            if (pass == 0) { Gh[(size_t)(nz + r) * Rs + c] = xn[i]; Gh[(size_t)c * Rs + nz + r] = xn[i]; }
                            else Gh[(size_t)(nz + r) * Rs + nz + c] = xn[i];
                        }
                    }
                }
            }
        }
        __threadfence();   // pass 1 reads what other threads of the workgroup wrote in pass 0 (and the diagonal below, pass 1)
        __syncthreads();
    }
    for (int r = tid; r < Rs; r += GHAT_THREADS) {
        const double v = r < R ? Gh[(size_t)r * Rs + r] : 1.0;
        gn[r] = sqrt(v > 0.0 ? v : 1.0);
    }
}

// Design-time table for PolishGenParams::s0_basis: block j < n rolls out (e0 = unit_j, v0 = V[:, j]), block n (e0 = 0, v0 = v0S) with
// the finish's own rollout and reads the state rows where the finish reads them.
struct S0BasisParams {
    int n, m, N, nz, nzs, R, Rs, roll_g, roll_cpl;
    const double* A; const double* B; const double* Vs; const double* v0S; const double* dvec; const int* row_traj;
    double* out;   // [(n + 1)][Rs], zero where no state row sits
};
inline __global__ __launch_bounds__(64) void k_s0_basis(S0BasisParams p) {
    extern __shared__ __attribute__((aligned(16))) double Z[];
    const int j = blockIdx.x, lane = threadIdx.x, n = p.n, m = p.m, N = p.N, C = n + m;
    for (int t = lane; t < (N + 1) * C; t += 64) Z[t] = 0.0;
    wave_fence_lds();
    for (int r = lane; r < p.nz; r += 64) {
        const double v = j < n ? p.Vs[(size_t)j * p.nzs + r] : (p.v0S ? p.v0S[r] : 0.0);
        Z[(size_t)(r / m) * C + n + r % m] = v * p.dvec[r];
    }
    if (lane < n) Z[lane] = (lane == j) ? 1.0 : 0.0;
    wave_fence_lds();
    switch (p.roll_cpl) {
        case 1: rollout_steps<1>(Z, n, m, N, p.roll_g, lane, p.A, p.B); break;
        case 2: rollout_steps<2>(Z, n, m, N, p.roll_g, lane, p.A, p.B); break;
        case 4: rollout_steps<4>(Z, n, m, N, p.roll_g, lane, p.A, p.B); break;
        default: rollout_steps<8>(Z, n, m, N, p.roll_g, lane, p.A, p.B); break;
    }
    for (int r = lane; r < p.Rs; r += 64) p.out[(size_t)j * p.Rs + r] = (r >= p.nz && r < p.R) ? Z[p.row_traj[r]] : 0.0;
}

// first launch: one wave per instance, working sets up to 32 rows (two waves per SIMD: at most 256 registers)
template <int NP>
__global__ __launch_bounds__(64 * PGEN_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_polish_gen(PolishGenParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    polish_gen_body<NP, 32>(p, smem);
}

// Helper wave w = 1..3 of the second launch's workgroup: waits for a job (k rows, instance), sums its share of
// sum_l Ghat[W_l, :] a_l (rows l = w, w + 4, ...; a in pbufb, W in wrow_s: the main wave's LDS) into its slice of `part`, and waits
// again; k < 0: the main wave is done.  Barrier for barrier the mirror of polish_gen_body<.., COOP>::g_rows_minus.
template <int NP>
__device__ __forceinline__ void pgen_helper(const PolishGenParams& p, double* smem, int w) {
    constexpr int WL = 64, CH = 8;
    const int lane = threadIdx.x & 63;
    const double* pbufb = smem + WL * WL + 512 + 512 + WL;
    const int* wrow_s = reinterpret_cast<const int*>(pbufb + WL);
    const int* job = reinterpret_cast<const int*>(smem + pgen_lds_per_wave(WL));
    double* part = w == 1 ? smem + WL * WL /* the main wave's rowbuf: free during a stream */ : smem + pgen_lds_per_wave(WL) + 2 + (size_t)(w - 2) * 512;
    const int Rs = p.Rs;
    for (;;) {
        __syncthreads();
        const int k = job[0];
        if (k < 0) return;
        const double* Gh = p.Ghat + (size_t)job[1] * p.Ghat_stride;
        double acc[NP][2];
#pragma unroll
        for (int q = 0; q < NP; ++q) { acc[q][0] = 0.0; acc[q][1] = 0.0; }
        constexpr int NW = PGEN_HELPERS + 1;
        for (int l0 = w; l0 < k; l0 += NW * CH) {
            d2 g[CH][NP];
            double av[CH];
#pragma unroll
            for (int t = 0; t < CH; ++t) {
                const int l = l0 + NW * t;
                const int lc = l < k ? l : 0;
                const double* row = Gh + (size_t)wrow_s[lc] * Rs;
                av[t] = l < k ? pbufb[lc] : 0.0;
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const int off = 2 * lane + 128 * q;
                    g[t][q] = *reinterpret_cast<const d2*>(row + (off < Rs ? off : 0));
                }
            }
#pragma unroll
            for (int t = 0; t < CH; ++t)
#pragma unroll
                for (int q = 0; q < NP; ++q) { acc[q][0] += g[t][q][0] * av[t]; acc[q][1] += g[t][q][1] * av[t]; }
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            d2 v; v[0] = acc[q][0]; v[1] = acc[q][1];
            *reinterpret_cast<d2*>(part + 2 * lane + 128 * q) = v;
        }
        __syncthreads();
    }
}
__host__ __device__ constexpr int pgen_coop_lds_doubles() { return pgen_lds_per_wave(64) + 2 + (PGEN_HELPERS > 1 ? PGEN_HELPERS - 1 : 0) * 512; }

// second launch: the instances the first one flagged, working sets up to 64 rows.  One instance per workgroup of FOUR waves: wave 0
// runs the finish, waves 1 - 3 take three quarters of every Ghat[W,:] stream -- with 250 - 400 flagged instances on 256 CUs the chip
// is idle otherwise, and a single wave's stream of 40 - 64 rows was half of its 13 k cycles per working-set change
template <int NP>
__global__ __launch_bounds__(64 * (PGEN_HELPERS + 1)) void k_polish_gen64(PolishGenParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int w = threadIdx.x >> 6;
    if (w == 0) {
        polish_gen_body<NP, 64, true>(p, smem);
        int* job = reinterpret_cast<int*>(smem + pgen_lds_per_wave(64));
        if ((threadIdx.x & 63) == 0) job[0] = -1;
        __syncthreads();
    } else pgen_helper<NP>(p, smem, w);
}

}  // namespace almpc
