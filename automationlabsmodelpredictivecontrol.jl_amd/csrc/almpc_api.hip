// almpc_api.hip -- C ABI of libalmpc.so (declared in include/almpc.h) over the gfx950 kernels.
//
// Host side of the drop-in boundary.  Reference entry points this replaces (relative to
// /root/reference): _model_predictive_control_design src/sub/design_mpc.jl:54-129 (-> almpc_design_shared),
// update_initialization! src/main/computation_mpc.jl:17-29, calculate! src/main/computation_mpc.jl:38-55.
// No torch, no BLAS, no CPU solve path: every per-step computation runs in almpc_kernels.hip.h.
#include "almpc_kernels.hip.h"
#include "almpc_design.hip.h"
#include "almpc_polish_gen.hip.h"
#include "almpc_fnn.hip.h"
#include "almpc_instance.hip.h"
#include "almpc_sqp.hip.h"
#include "almpc_comm.hip.h"
#include "almpc_riccati.hip.h"
#include "almpc_sdual.hip.h"
#include "almpc_host_math.h"
#include "../../include/almpc.h"

// The heavy kernel templates are compiled by their own translation units (almpc_tu_*.hip, one per kernel family, built in parallel);
// here their instantiations are only DECLARED.  The lists are generated from a unity build (tools/gen_instances.py); an instantiation
// that is missing from them is simply compiled here, as everything is with -DALMPC_UNITY (one translation unit: the diagnostic
// -DALMPC_STAMPS build, whose stamp buffer is a device variable of ONE code object).
#ifndef ALMPC_UNITY
#define ALMPC_KERNEL_INSTANCE(...) extern template __global__ __VA_ARGS__;
#include "instances/sdual_a.inc"
#include "instances/sdual_b.inc"
#include "instances/sdual_c.inc"
#include "instances/polish_gen.inc"
#include "instances/step.inc"
#include "instances/instance.inc"
#include "instances/design_a.inc"
#include "instances/design_b.inc"
#undef ALMPC_KERNEL_INSTANCE
#endif

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

using namespace almpc;

struct almpc_handle {
    int n = 0, m = 0, N = 0, batch = 0, nz = 0, nzs = 0, nrb = 0, ks = 0, ksf = 0, device = 0;
    uint32_t flags = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool designed = false;
    double rho = 0.1, sigma = 1e-6;
    // host copies of the design (almpc_get_design)
    std::vector<double> H, F, P, d;
    // device: shared design
    double *dMinvFrag = nullptr, *dVFrag = nullptr, *dHFrag = nullptr, *dFFrag = nullptr, *dG = nullptr;
    double *dD = nullptr, *dUmin = nullptr, *dUmax = nullptr, *dA = nullptr, *dB = nullptr;
    double *dXref = nullptr, *dUref = nullptr, *dFS = nullptr, *dV0S = nullptr, *dRho = nullptr;
    size_t ref_cap[3] = {0, 0, 0};   // doubles in dXref, dUref, dFS (= dV0S) as almpc_set_reference allocated them: re-used while the sizes stay
    double *wQ = nullptr, *wR = nullptr, *wS = nullptr;   // design weights on the device (almpc_design_batched / _ltv)
    int rho_mode = 0;  // 0 scalar rho (OSQP), 1 stiffness profile rho / G_ii
    // blocked rollout of the shared model (rollout_blocked): [Gamma_s | Phi_s] rows per lane, built at design time
    double* dRollM = nullptr;
    int roll_s = 0, roll_nb = 0;  // roll_s == 0: shape not covered (n > ROLL_NX or m > ROLL_SMX), the stage-by-stage rollout is used
    long xref_stride = 0, uref_stride = 0, fS_stride = 0;
    std::vector<double> hS;  // S weight, symmetrised (for fS with per-instance references)
    int useS = 0;            // the input-rate term is part of the cost: R[1,1] != 0 and S[1,1] != 0 (src/sub/design_mpc.jl:423-466)
    // device: per-instance state and results
    double *dX0 = nullptr, *dXs = nullptr, *dZs = nullptr, *dYs = nullptr, *dV0 = nullptr, *dW = nullptr;
    double *dX = nullptr, *dEx = nullptr, *dU = nullptr, *dEu = nullptr;
    int32_t *dStatus = nullptr, *dIters = nullptr, *dPiters = nullptr, *dPerm = nullptr;
    uint32_t* dYflags = nullptr;  // [batch][nrb] signs of the ADMM multipliers (steps run with ALMPC_OPT_NO_WARM_STATE)
    bool state_valid = true;      // xs / ys hold the ADMM state of the last step (a warm start may use them)
    int num_cus = 256;            // persistent-grid size of k_polish<true>
    int polish_no_glds = 0;       // diagnostic: force the G-through-L2 build (ALMPC_POLISH_NO_GLDS=1)
    int skip_admm = 0;            // transient (SQP iterations after the first): guess from the iterate, no ADMM phase, no KKT inverse;
                                  // 2 (re-linearisation pipeline, warm steps): guess = the previous step's inputs shifted by one stage
    int fuse_step = 1;            // one kernel per step when the shape allows (almpc_set_step_fusion / ALMPC_NO_FUSED_STEP=1)
    double* dSglobal = nullptr;  // polish scratch for working sets beyond 32 rows
    int32_t* dStartRows = nullptr;   // [batch][65] row list of a guessed working set whose inverse sits in dSglobal (k_guess_iterate_ws)
    // state rows (state box / terminal equality): constraint-space data for k_polish_gen
    int terminal_eq = 0, has_box = 0, mc = 0, R = 0, Rs = 0, np_pairs = 0;
    double *dGhat = nullptr, *dGnorm = nullptr, *dXmin = nullptr, *dXmax = nullptr;
    int *dRowTraj = nullptr, *dRowEq = nullptr, *dRowXidx = nullptr, *dRowState = nullptr;
    double* dOvfSinv = nullptr;    // [batch][32 * 32 + 32] k_polish_gen -> k_polish_gen64: inverse and bounds of a flagged instance
    double* dVsPlain = nullptr;    // [n][nzs] V = -G F' (shared design with state rows): operand of the s0 table
    double* dPlain = nullptr;      // small shared designs (nzs <= 64): dense [Minv | H' | F' | V] for the one-wave-per-instance step
    double* dS0Basis = nullptr;    // [(n + 1)][Rs] PolishGenParams::s0_basis
    bool s0_basis_ok = false;
    int32_t* dOverflow = nullptr;  // [2 + batch] k_polish_gen: count, cursor and list of instances to redo with the 64-row build
    int* dRowMap = nullptr;        // [N*n] state (stage k+2, i) -> state-row index or -1 (k_ghat_inst)
    double *dGhatE = nullptr, *dWinvE = nullptr;  // shared design with the terminal equality: original rows E of Ghat, Ghat_EE^-1
    int eq_proj = 0;               // dGhat is the matrix projected on the terminal equality (k_ghat_project)
    bool ghat_inst = false;        // dGhat / dGnorm hold one constraint-space matrix PER INSTANCE ([batch][R][Rs], [batch][Rs])
    std::vector<double> boxmin, boxmax;  // almpc_set_state_box: the state box of the per-instance / time-varying / SQP designs
    double *lA = nullptr, *lB = nullptr, *lC = nullptr, *lE = nullptr;  // almpc_design_ltv with state rows: stage models, defects and
                                                                        // state errors kept for the step's rollouts
    // per-instance models (almpc_design_batched): persistent per-instance operands ...
    bool batched = false;
    bool ltv = false;             // almpc_design_ltv: references and gradient are part of the design
    double* bQ = nullptr;         // [batch][nz] explicit gradient of an LTV design (unscaled)
    double *bA = nullptr, *bB = nullptr, *bMinv = nullptr, *bG = nullptr, *bHs = nullptr, *bFs = nullptr, *bVs = nullptr,
           *bD = nullptr, *bRho = nullptr, *bH = nullptr, *bF = nullptr;
    // ... and design temporaries kept for the next re-design (a per-step re-linearisation designs every step)
    double *bPhi = nullptr, *bGk = nullptr, *bGam = nullptr, *bW = nullptr, *bWP = nullptr, *bP = nullptr;
    int* bFlag = nullptr;
    bool batched_alloc = false;
    bool minv_packed = false;   // bMinv holds packed lower triangles (stride packed_tri_doubles): k_admm_inst<true>
    // SQP outer loop for a black-box Fnn model (almpc_sqp_fnn_*): the network, the stage data of the current linearisation
    struct Sqp {
        bool ready = false, started = false;
        int H = 0, L = 0, act = 0, useR = 0, useS = 0;
        long sP = 0;
        double *W_in = nullptr, *W_h = nullptr, *b_h = nullptr, *W_out = nullptr;
        double *A = nullptr, *B = nullptr, *c = nullptr, *fval = nullptr, *ebar = nullptr, *qadd = nullptr;
        double *xref = nullptr, *uref = nullptr, *Q = nullptr, *R = nullptr, *S = nullptr;
        int* bad = nullptr;
        double* mer = nullptr;    // [batch][4] step rule 1: step factor, merit of the last accepted point, redo flag
        double *xback = nullptr, *uback = nullptr, *dxback = nullptr, *vback = nullptr;  // last accepted point and its step
        double mu = 0.0;          // merit weight of the defects
        int step_rule = 0;        // 0 fixed step, 1 merit-function safeguard (almpc_sqp_fnn_set_step_rule)
        long since_start = 0;     // iterations since almpc_sqp_fnn_start: the first one gets its guess from ADMM, the others from the iterate
        int guess_from_iterate = 1;  // diagnostic: ALMPC_SQP_ADMM_ALWAYS=1 keeps the ADMM phase in every iteration
        int structured_qp = 0;       // almpc_sqp_fnn_set_structured: every iteration's QP goes to k_riccati in its stage-wise form
                                     // (no condensed design at all: no Hessian build, no inverse, no m N <= 128 limit)
        unsigned long long* stats = nullptr;  // [iters][2]
        int stats_cap = 0;
    } sqp;
    // per-step re-linearisation of a black-box Fnn model on the device (almpc_relin_fnn_*, BASELINE configs[3])
    struct Relin {
        bool ready = false;
        int H = 0, L = 0, act = 0, useR = 0, useS = 0;
        double *W_in = nullptr, *W_h = nullptr, *b_h = nullptr, *W_out = nullptr;
        double *ulin = nullptr;   // [batch][m] linearisation input of every instance (the first input reference)
        double *Q = nullptr, *R = nullptr, *S = nullptr;
        double *gS = nullptr;     // [nz] unscaled input-rate gradient 2 D'Sbar D u_ref of the shared reference
        bool have_prev = false;   // a step has been solved since setup: its inputs can seed the next step's working set
        double *u0 = nullptr, *xnext = nullptr;   // [batch][m] applied inputs, [batch][n] next states (almpc_relin_fnn_advance)
        double *Ascr = nullptr, *Bscr = nullptr;  // Jacobian outputs of the advance's forward pass (not used)
        float ms_jac = 0, ms_design = 0, ms_step = 0;  // last timed step (almpc_relin_fnn_step with timing)
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    } relin;
    // structured (Riccati) solve: the handle's only solver (ALMPC_FLAG_STRUCTURED) or the fallback for instances the condensed path
    // leaves unsolved (almpc_set_structured_fallback)
    bool structured = false;
    // lazy redo (fallback == 2 on the shared-model, input-box-only path, whose steps are tens of microseconds and leave an instance
    // unsolved only in corner cases): the finish counts such instances into a host-visible word; the redo kernels are launched at the
    // next host sync point only when the count has moved -- no extra launch on the step path (measured: two idle redo launches cost
    // 14 us per 61 us step)
    int* hUnsolved = nullptr;      // pinned host word
    int* dUnsolved = nullptr;      // the device's address of it
    int unsolved_seen = 0;
    bool lazy_pending = false;
    bool redo_x0_from_results = false;   // set around the launches of a lazy redo: x0 = stage 1 of the step's own x (the caller may have
                                         // handed over the next x0 since)
    // Gated redo (round 5): where results leave without a synchronous host look -- the asynchronous tickets (almpc_get_results_async)
    // and the device-resident loops (almpc_advance_plant, almpc_relin_fnn_advance) -- the redo launches are enqueued behind the step
    // with a GATE: the finish of step s stores s in dRedoGate when it leaves an instance undecided, and a gated launch returns at once
    // unless the word holds its step's number.  Every read path hands out "solution or verdict" (src/main/computation_mpc.jl:41-53)
    // at the cost of two (three with per-instance stage records) empty launches per step on those paths only.
    int* dRedoGate = nullptr;
    int step_serial = 0;                 // number of the last enqueued step (1, 2, ...)
    bool redo_gate_on = false;           // set around the launches of a gated redo
    bool redo_expected = false;          // the last synchronous look found undecided instances (see wait_and_settle)
    bool redo_predicted = false;         // set around the launches of a gated redo that is EXPECTED to have work (tier policy of a plain redo)
    bool flag_in_finish = false;   // transient (re-linearisation step): the finish turns a flagged design into ALMPC_NON_FINITE itself
    int fallback = 2;   // 0 off, 1 asked for (a design it cannot serve is an error), 2 default: on wherever the stage-wise solvers cover the design
    double *rQ = nullptr, *rR = nullptr, *rP = nullptr, *rKst = nullptr, *rPst = nullptr;   // device copies of Q, R (branch rule applied), shared P; gain scratch
    double* rGuess = nullptr;   // [batch][N][m] start of the next structured solve (almpc_set_start_from / opts.warm_start), else nullptr
    bool guess_ready = false;   // rGuess was filled for the NEXT almpc_calculate (consumed by it)
    bool r_has_step = false;    // a structured step has run on this design: its inputs can seed a warm start
    hipEvent_t ev_guess = nullptr, ev_guess_done = nullptr;
    long rP_stride = 0;   // per-instance terminal weights (batched designs): doubles between instances of bP, else 0 with rP
    bool r_batched_P = false;
    // stage-wise dual active-set solve (k_sdual, csrc/almpc_sdual.hip.h): input box, state box, terminal equality and S in the
    // multiple-shooting form; stage records of the unconstrained problem (shared: host Riccati at design time)
    struct Sd {
        bool ready = false;
        int nt = 0, NT = 0, MC = 0;          // stage-state dimension (n, or n + m with S) and the instantiated (padded) dimensions
        double* rec = nullptr; size_t rec_cap = 0; long rec_stride = 0, rec_kstride = 0;
        double* base = nullptr; size_t base_cap = 0; long base_stride = 0; bool has_base = false;
        double *xmin = nullptr, *xmax = nullptr, *eqt = nullptr;   // state box [n] (null: none), terminal-equality target [n] zeros (null: none)
        bool has_box = false, has_eq = false, useS = false;
        bool per_instance = false;          // records per instance (k_sgains) instead of the host's shared ones
        int gain_N = 0;                     // stages k_sgains computes (1: stage-invariant records)
        double *dQ = nullptr, *dR = nullptr, *dS = nullptr, *dP = nullptr;   // weights of k_sgains (R with the branch rule applied; dP: a shared terminal weight)
        int* bad = nullptr;                 // [batch] k_sgains: R + B'PB not positive definite
        bool sqp = false;                   // the QP of an SQP iteration: stage models, defects and cost terms of the loop (h->sqp)
        double *pc = nullptr, *ct = nullptr;   // [batch][N][NT] P_{k+1} c_k, c_k
        int32_t* ovf = nullptr;
        int32_t* wsave = nullptr;   // [batch][SDUAL_WSAVE] working set of an instance that ran out of room (start of the next tier)
        double* sinv_save = nullptr;   // [batch][sdual_sinv_doubles(SDUAL_SINV_SAVE)] its inverse (allocated at the first multi-tier solve)
        int tier_serial = 0;           // number of the last launch_sdual_t call (the tiers' gate value)
        double* start_inv = nullptr;   // [batch][sdual_sinv_doubles(SDUAL_SINV_SAVE)] inverse of a redo's start (k_sdual_start; allocated at the first such redo)
        int32_t* start_ws = nullptr;   // [batch][64] working sets the state-row finish of the LAST step gave up with (PolishGenParams::redo_ws)
        bool start_ws_fresh = false;   // ... written by the last enqueued step (cleared by every step that does not run that finish)
        double* sinv_glb = nullptr; size_t sinv_cap = 0;   // third tier: Sinv of 128 x 129 per wave of its grid
        double* ghat = nullptr; size_t ghat_cap = 0; bool ghat_ready = false, ghat_building = false, ghat_wanted = false;   // shared model: cached sweep responses [TP][TP] (k_sdual: SdualParams::ghat)
        std::vector<double> S;               // symmetrised S (base terms of time-varying input references)
        // reachability screen of the state box (k_state_box_screen): tables of the shared model and references, verdicts per instance
        double *scr_phi = nullptr, *scr_g = nullptr, *scr_rm = nullptr, *scr_rp = nullptr;
        int32_t* scr_verdict = nullptr;
        size_t scr_cap = 0;                  // N the table buffers were sized for
        bool scr_ready = false;              // tables match the current model / references
    } sd;
    // multi-GPU (almpc_comm_*): this handle's rank in an RCCL communicator of one process per GPU
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    long long* dComm4 = nullptr;       // [4] summary words
    double *dU0 = nullptr, *dU0all = nullptr;  // [batch][m] packed first inputs, [world][batch][m] gathered
    // host-facing step path (almpc_update_initialization_async, almpc_get_results_async / _wait, almpc_get_first_input): pinned
    // staging owned by the handle, copies on their own streams, rings of IO_DEPTH slots so that the transfers of one step run
    // under the kernel of the next (csrc/almpc_hostio.inc.h)
    static constexpr int IO_DEPTH = 2;
    struct Io {
        bool ready = false;
        hipStream_t s_out = nullptr;
        // ALMPC_X0_UPLOAD=1: the pinned slot is copied into a device slot by the copy engine on a stream of its own, under the step that
        // is running, instead of being read in place by the kernels.  Reading in place costs the step +9 us at the benchmark shape
        // (393 KB over the link at its START, tools/dbg_x0_home.py), the upload costs three HIP calls and a copy-engine latency per step:
        // measured 11.7 k (upload) against 12.8 k (in place) batch-steps/s on the pipelined first-move loop, 8.3 k against 10.4 k serial
        hipStream_t s_in = nullptr;
        bool upload = false;
        double* dX0dev[IO_DEPTH] = {nullptr, nullptr};
        hipEvent_t ev_in[IO_DEPTH] = {nullptr, nullptr};
        // x0 ring: pinned host slots the kernels read in place (dX0 = the device's address of the slot); h->dX0 points at the latest
        double* hX0[IO_DEPTH] = {nullptr, nullptr};
        double* dX0[IO_DEPTH] = {nullptr, nullptr};
        hipEvent_t ev_used[IO_DEPTH] = {nullptr, nullptr};   // the last step that read the slot has finished (recorded on the compute stream)
        bool used_pending[IO_DEPTH] = {false, false};
        int x0_slot = -1;     // slot h->dX0 points at (-1: the handle's own buffer)
        double* dX0_own = nullptr;   // the handle's own x0 buffer while h->dX0 points into the ring
        long x0_count = 0;
        // result ring: ticket t lives in slot t % IO_DEPTH
        double *hX[IO_DEPTH] = {nullptr, nullptr}, *hEx[IO_DEPTH] = {nullptr, nullptr}, *hU[IO_DEPTH] = {nullptr, nullptr},
               *hEu[IO_DEPTH] = {nullptr, nullptr}, *hU0[IO_DEPTH] = {nullptr, nullptr};
        int32_t* hInts[IO_DEPTH] = {nullptr, nullptr};       // pinned [3][batch]: status | iters | polish_iters
        double* dU0[IO_DEPTH] = {nullptr, nullptr};          // the device's addresses of hU0 / hInts (the pack kernel writes the pinned
        int32_t* dInts[IO_DEPTH] = {nullptr, nullptr};       // slots directly)
        hipEvent_t ev_packed[IO_DEPTH] = {nullptr, nullptr}; // compute stream: the results of the slot's step exist (copy-out stream waits)
        hipEvent_t ev_done[IO_DEPTH] = {nullptr, nullptr};   // everything the slot's request asked for has landed in pinned memory
        hipEvent_t ev_big = nullptr;                          // the last read-back of x / e_x / u / e_u has left the result buffers (the next step waits for it)
        uint32_t want[IO_DEPTH] = {0, 0};
        long ticket[IO_DEPTH] = {-1, -1};
        long next_ticket = 0;
        bool big_copy_pending = false;  // a read-back of x / e_x / u / e_u (straight from the result buffers) may still be running:
        int big_copy_slot = 0;          // the next step waits for it before it overwrites them
    } io;
    // timing (ALMPC_FLAG_TIMING): one set of 4 events per step since the last almpc_timing_reset
    std::vector<hipEvent_t> ev;  // 4 per step
    std::vector<char> ev_two;    // per recorded step: 1 = a one-kernel step, only its events 1 and 2 (around the kernel) were recorded
    size_t ev_used = 0;          // steps recorded
    int timing_stride = 1;       // record events on every timing_stride-th step only (each event costs ~3 us of stream time)
    size_t step_count = 0;
};

namespace {

int fail(almpc_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    return code;
}

#define HIP_TRY(h, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(h, ALMPC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));   \
    } while (0)

template <typename T>
hipError_t dalloc(T** p, size_t count) {
    return hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T) + 64);
}

int pick_ks(int nz, int nrb) {
    const int exact = (nz + 3) / 4;
    // instantiated (NRB, KS) pairs: KS = 4*NRB always; plus the exact-fit specials below
    if (nrb == 8 && exact <= 30) return 30;
    if (nrb == 3 && exact <= 10) return 10;
    if (nrb == 1 && exact <= 3) return 3;
    return 4 * nrb;
}

template <int NRB, int KS>
hipError_t launch_admm_t(const AdmmParams& p, int grid, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((k_admm<NRB, KS>), dim3(grid), dim3(64 * NRB), lds, st, p);
    return hipGetLastError();
}

hipError_t launch_admm(int nrb, int ks, const AdmmParams& p, int grid, size_t lds, hipStream_t st) {
#define CASE(NRB_, KS_) if (nrb == NRB_ && ks == KS_) return launch_admm_t<NRB_, KS_>(p, grid, lds, st)
    CASE(1, 3); CASE(1, 4); CASE(2, 8); CASE(3, 10); CASE(3, 12); CASE(4, 16); CASE(5, 20); CASE(6, 24);
    CASE(7, 28); CASE(8, 30); CASE(8, 32);
#undef CASE
    return hipErrorInvalidValue;
}

void io_free(almpc_handle* h);   // almpc_hostio.inc.h

// After almpc_update_initialization_async h->dX0 points at a pinned host slot the kernels read in place.  Entry points that WRITE x0
// through a device-side copy go back to the handle's own device buffer first.
void io_release_x0(almpc_handle* h) {
    if (h->io.x0_slot >= 0) {
        h->dX0 = h->io.dX0_own;
        h->io.x0_slot = -1;
    }
}

void free_all(almpc_handle* h) {
    io_free(h);
    void* ptrs[] = {h->dMinvFrag, h->dVFrag, h->dHFrag, h->dFFrag, h->dG, h->dD, h->dUmin, h->dUmax, h->dA, h->dB,
                    h->dXref, h->dUref, h->dFS, h->dV0S, h->dRho, h->dRollM, h->dX0, h->dXs, h->dZs, h->dYs, h->dV0, h->dW, h->dX, h->dEx,
                    h->dU, h->dEu, h->dStatus, h->dIters, h->dPiters, h->dPerm, h->dYflags, h->dSglobal, h->dStartRows, h->dGhat, h->dGnorm, h->dXmin,
                    h->dXmax, h->dRowTraj, h->dRowEq, h->dRowXidx, h->dRowState, h->dRowMap, h->dGhatE, h->dWinvE, h->lA, h->lB, h->lC, h->lE, h->bA, h->bB, h->bMinv, h->bG, h->bHs, h->bFs,
                    h->bVs, h->bD, h->bRho, h->bH, h->bF, h->bPhi, h->bGk, h->bGam, h->bW, h->bWP, h->bP, h->bFlag, h->bQ, h->dOverflow, h->dOvfSinv, h->dVsPlain, h->dPlain, h->dS0Basis, h->wQ, h->wR, h->wS,
                    h->sqp.W_in, h->sqp.W_h, h->sqp.b_h, h->sqp.W_out, h->sqp.A, h->sqp.B, h->sqp.c, h->sqp.fval, h->sqp.ebar,
                    h->sqp.qadd, h->sqp.xref, h->sqp.uref, h->sqp.Q, h->sqp.R, h->sqp.S, h->sqp.bad, h->sqp.stats, h->sqp.mer, h->sqp.xback, h->sqp.uback, h->sqp.dxback, h->sqp.vback};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (void* p : {(void*)h->relin.W_in, (void*)h->relin.W_h, (void*)h->relin.b_h, (void*)h->relin.W_out, (void*)h->relin.ulin,
                    (void*)h->relin.Q, (void*)h->relin.R, (void*)h->relin.S, (void*)h->relin.gS, (void*)h->relin.u0, (void*)h->relin.xnext, (void*)h->relin.Ascr, (void*)h->relin.Bscr})
        if (p) (void)hipFree(p);
    for (auto& e : h->relin.ev)
        if (e) (void)hipEventDestroy(e);
    if (h->comm && rccl_api().ok) (void)rccl_api().CommDestroy(h->comm);
    for (void* p : {(void*)h->dComm4, (void*)h->dU0, (void*)h->dU0all, (void*)h->rQ, (void*)h->rR, (void*)h->rP, (void*)h->rKst, (void*)h->rPst, (void*)h->rGuess,
                    (void*)h->sd.rec, (void*)h->sd.base, (void*)h->sd.xmin, (void*)h->sd.xmax, (void*)h->sd.eqt, (void*)h->sd.ovf, (void*)h->sd.wsave, (void*)h->sd.sinv_glb, (void*)h->sd.ghat, (void*)h->sd.dQ, (void*)h->sd.dR, (void*)h->sd.dS, (void*)h->sd.dP, (void*)h->sd.bad, (void*)h->sd.pc, (void*)h->sd.ct,
                    (void*)h->sd.sinv_save, (void*)h->sd.start_inv, (void*)h->sd.start_ws, (void*)h->sd.scr_phi, (void*)h->sd.scr_g, (void*)h->sd.scr_rm, (void*)h->sd.scr_rp, (void*)h->sd.scr_verdict})
        if (p) (void)hipFree(p);
    if (h->ev_guess) (void)hipEventDestroy(h->ev_guess);
    if (h->ev_guess_done) (void)hipEventDestroy(h->ev_guess_done);
    if (h->hUnsolved) (void)hipHostFree(h->hUnsolved);
    if (h->dRedoGate) (void)hipFree(h->dRedoGate);
    for (auto& e : h->ev)
        if (e) (void)hipEventDestroy(e);
    h->ev.clear();
    if (h->stream) (void)hipStreamDestroy(h->stream);
}

// State rows of a design: the state box for stages 2..N+1 (stage 1 is x0 itself, checked per instance) and / or the terminal
// equality on stage N+1 (which then replaces the box rows of that stage).  Fills the row tables and (re)allocates the
// constraint-space matrix: one for the handle (shared model) or one per instance.  mc = 0 afterwards: no state rows.
int setup_state_rows(almpc_handle* h, const double* xmin, const double* xmax, bool per_instance) {
    const int n = h->n, m = h->m, N = h->N, nz = h->nz;
    h->has_box = xmin ? 1 : 0;
    std::vector<int> row_traj, row_eq, row_xidx, row_state, rowmap((size_t)N * n, -1);
    if (h->has_box)
        for (int i = 0; i < n; ++i)
            if (!(xmin[i] <= xmax[i])) return fail(h, ALMPC_ERR_INVALID, "design: xmin > xmax");
    for (int k = 0; k < N; ++k)
        for (int i = 0; i < n; ++i) {
            const bool is_eq = h->terminal_eq && k == N - 1;
            if (!(h->has_box || is_eq)) continue;
            rowmap[(size_t)k * n + i] = (int)row_traj.size();
            row_traj.push_back((k + 1) * (n + m) + i);
            row_eq.push_back(is_eq ? 1 : 0);
            row_xidx.push_back((k + 1) * n + i);
            row_state.push_back(i);
        }
    h->mc = (int)row_traj.size();
    h->R = nz + h->mc;
    h->np_pairs = (h->R + 127) / 128;
    h->Rs = 128 * h->np_pairs;
    for (void* q : {(void*)h->dGhat, (void*)h->dGnorm, (void*)h->dXmin, (void*)h->dXmax, (void*)h->dRowTraj, (void*)h->dRowEq,
                    (void*)h->dRowXidx, (void*)h->dRowState, (void*)h->dRowMap})
        if (q) (void)hipFree(q);
    h->dGhat = h->dGnorm = h->dXmin = h->dXmax = nullptr;
    h->dRowTraj = h->dRowEq = h->dRowXidx = h->dRowState = h->dRowMap = nullptr;
    h->ghat_inst = false;
    if (h->dGhatE) { (void)hipFree(h->dGhatE); h->dGhatE = nullptr; }
    if (h->dWinvE) { (void)hipFree(h->dWinvE); h->dWinvE = nullptr; }
    h->eq_proj = 0;
    if (h->mc == 0) return ALMPC_OK;
    if (h->np_pairs > 4) return fail(h, ALMPC_ERR_UNSUPPORTED, "design: state rows need n*N + m*N <= 512");
    if ((size_t)(N + 1) * (n + m) > 32 * 32) return fail(h, ALMPC_ERR_UNSUPPORTED, "design: state rows need (N+1)*(n+m) <= 1024");
    if (per_instance && n > 32) return fail(h, ALMPC_ERR_UNSUPPORTED, "design: state rows with per-instance models need n <= 32");
    const size_t copies = per_instance ? (size_t)h->batch : 1;
    HIP_TRY(h, dalloc(&h->dGhat, copies * h->R * h->Rs));
    HIP_TRY(h, dalloc(&h->dGnorm, copies * h->Rs));
    if (per_instance) HIP_TRY(h, hipMemset(h->dGhat, 0, copies * h->R * h->Rs * sizeof(double)));  // (the padding columns stay zero)
    HIP_TRY(h, dalloc(&h->dXmin, (size_t)n)); HIP_TRY(h, dalloc(&h->dXmax, (size_t)n));
    HIP_TRY(h, dalloc(&h->dRowTraj, (size_t)h->Rs)); HIP_TRY(h, dalloc(&h->dRowEq, (size_t)h->Rs));
    HIP_TRY(h, dalloc(&h->dRowXidx, (size_t)h->Rs)); HIP_TRY(h, dalloc(&h->dRowState, (size_t)h->Rs));
    HIP_TRY(h, dalloc(&h->dRowMap, rowmap.size()));
    HIP_TRY(h, hipMemcpy(h->dRowMap, rowmap.data(), rowmap.size() * sizeof(int), hipMemcpyHostToDevice));
    auto up = [&](int* dst, const std::vector<int>& v, int fill) {
        std::vector<int> full((size_t)h->Rs, fill);
        for (size_t i = 0; i < v.size(); ++i) full[(size_t)nz + i] = v[i];
        return hipMemcpy(dst, full.data(), full.size() * sizeof(int), hipMemcpyHostToDevice);
    };
    HIP_TRY(h, up(h->dRowTraj, row_traj, 0)); HIP_TRY(h, up(h->dRowEq, row_eq, 0));
    HIP_TRY(h, up(h->dRowXidx, row_xidx, 0)); HIP_TRY(h, up(h->dRowState, row_state, 0));
    std::vector<double> lo(n, -1e300), hi(n, 1e300);
    if (h->has_box) { lo.assign(xmin, xmin + n); hi.assign(xmax, xmax + n); }
    HIP_TRY(h, hipMemcpy(h->dXmin, lo.data(), n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->dXmax, hi.data(), n * sizeof(double), hipMemcpyHostToDevice));
    h->ghat_inst = per_instance;
    return ALMPC_OK;
}

// Ghat_i of every instance from the per-instance design in place (G_i, d_i) and the models the rows roll out through: the
// time-invariant slots (bA, bB) or stage models [batch][N][..] (A_all / B_all non-null).  Launch only (handle's stream).
hipError_t launch_ghat_inst(almpc_handle* h, const double* A_all, const double* B_all) {
    GhatInstParams gp;
    gp.n = h->n; gp.m = h->m; gp.N = h->N; gp.nz = h->nz; gp.nzs = h->nzs; gp.mc = h->mc; gp.R = h->R; gp.Rs = h->Rs;
    const long n = h->n, m = h->m, N = h->N;
    if (A_all) { gp.A = A_all; gp.B = B_all; gp.A_stride = N * n * n; gp.B_stride = N * n * m; gp.A_kstride = n * n; gp.B_kstride = n * m; }
    else { gp.A = h->bA; gp.B = h->bB; gp.A_stride = n * n; gp.B_stride = n * m; gp.A_kstride = 0; gp.B_kstride = 0; }
    gp.G = h->bG; gp.G_stride = (long)h->nz * h->nzs; gp.dvec = h->bD; gp.d_stride = h->nzs;
    gp.rowmap = h->dRowMap; gp.Ghat = h->dGhat; gp.Ghat_stride = (long)h->R * h->Rs; gp.gnorm = h->dGnorm; gp.gnorm_stride = h->Rs;
    const size_t lds = (size_t)(n * (n + m) + 2) * sizeof(double);
    const dim3 grid((unsigned)h->batch), block(GHAT_THREADS);
    if (n <= 4) hipLaunchKernelGGL((k_ghat_inst<4>), grid, block, lds, h->stream, gp);
    else if (n <= 8) hipLaunchKernelGGL((k_ghat_inst<8>), grid, block, lds, h->stream, gp);
    else if (n <= 16) hipLaunchKernelGGL((k_ghat_inst<16>), grid, block, lds, h->stream, gp);
    else hipLaunchKernelGGL((k_ghat_inst<32>), grid, block, lds, h->stream, gp);
    return hipGetLastError();
}

// Device copies of the weights the structured solve uses (R with the reference's branch rule applied: zero if R[1,1] == 0)
int riccati_weights(almpc_handle* h, const hm::mat& Qm, const hm::mat& Rm, const double* Pshared) {
    const int n = h->n, m = h->m;
    hm::mat Qs = Qm, Rs = Rm;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < j; ++i) { const double v = 0.5 * (Qs[(size_t)j * n + i] + Qs[(size_t)i * n + j]); Qs[(size_t)j * n + i] = Qs[(size_t)i * n + j] = v; }
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < j; ++i) { const double v = 0.5 * (Rs[(size_t)j * m + i] + Rs[(size_t)i * m + j]); Rs[(size_t)j * m + i] = Rs[(size_t)i * m + j] = v; }
    if (Rm[0] == 0.0) std::fill(Rs.begin(), Rs.end(), 0.0);
    if (!h->rQ) HIP_TRY(h, dalloc(&h->rQ, (size_t)n * n));
    if (!h->rR) HIP_TRY(h, dalloc(&h->rR, (size_t)m * m));
    HIP_TRY(h, hipMemcpy(h->rQ, Qs.data(), Qs.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->rR, Rs.data(), Rs.size() * sizeof(double), hipMemcpyHostToDevice));
    if (Pshared) {
        if (!h->rP) HIP_TRY(h, dalloc(&h->rP, (size_t)n * n));
        HIP_TRY(h, hipMemcpy(h->rP, Pshared, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice));
    }
    if (!h->rKst) HIP_TRY(h, dalloc(&h->rKst, (size_t)h->batch * h->N * ((size_t)n * m + m)));
    if (!h->rPst && (size_t)h->batch * h->N * ((size_t)n * n + n) * sizeof(double) <= ((size_t)8 << 30))   // value functions of the last sweep
        HIP_TRY(h, dalloc(&h->rPst, (size_t)h->batch * h->N * ((size_t)n * n + n)));
    return ALMPC_OK;
}

bool riccati_shape_ok(const almpc_handle* h) {
    return h->n <= 32 && h->m <= 16 && (long)h->m * h->N <= 1024 && riccati_lds_doubles(h->n, h->m, h->N) * sizeof(double) <= 160 * 1024;
}

// k_riccati over the batch (filter = 0) or over the instances whose status is not 0 (filter = 1, start = the step's own result)
hipError_t launch_riccati(almpc_handle* h, int filter, const double* guess, int max_iter) {
    RiccatiParams rp;
    rp.n = h->n; rp.m = h->m; rp.N = h->N; rp.batch = h->batch;
    const bool pi = h->batched && !h->structured ? true : h->batched;
    rp.A = pi ? h->bA : h->dA; rp.A_stride = pi ? (long)h->n * h->n : 0;
    rp.B = pi ? h->bB : h->dB; rp.B_stride = pi ? (long)h->n * h->m : 0;
    rp.Q = h->rQ; rp.R = h->rR;
    rp.P = h->r_batched_P ? h->bP : h->rP; rp.P_stride = h->r_batched_P ? h->rP_stride : 0;
    if (!rp.Q || !rp.R || !rp.P || !h->rKst) return hipErrorInvalidValue;   // no riccati_weights() for this design: nothing to launch with
    rp.umin = h->dUmin; rp.umax = h->dUmax;
    rp.uref = h->dUref; rp.uref_stride = h->uref_stride; rp.xref = h->dXref; rp.xref_stride = h->xref_stride;
    rp.x0 = h->dX0; rp.x0_stride = h->n; rp.uguess = guess; rp.filter = filter; rp.Kst = h->rKst; rp.Pst = h->rPst;
    if (h->redo_x0_from_results) { rp.x0 = h->dX; rp.x0_stride = (long)h->n * (h->N + 1); }   // (lazy redo: see resolve_lazy_redo)
    if (h->redo_gate_on) { rp.gate = h->dRedoGate; rp.gate_val = h->step_serial; }
    rp.x = h->dX; rp.ex = h->dEx; rp.u = h->dU; rp.eu = h->dEu; rp.status = h->dStatus; rp.piters = h->dPiters;
    rp.max_iter = max_iter > 0 ? max_iter : 20 * h->N * h->m + 50;
    rp.tol = 1e-9;
    rp.lds_per_wave = riccati_lds_doubles(h->n, h->m, h->N);
    rp.A_kstride = 0; rp.B_kstride = 0; rp.c = nullptr; rp.c_stride = 0; rp.ebar = nullptr; rp.ebar_stride = 0;
    rp.qu = nullptr; rp.qu_stride = 0; rp.qu_scale = 1.0; rp.flag = nullptr; rp.v_only = 0;
    if (h->sqp.ready && filter >= 2) {   // the QP of the current SQP iteration: stage models, defects, state errors, input gradient
                                         // (2: instances the condensed path flagged or left unsolved; 3: every instance)
        const almpc_handle::Sqp& q = h->sqp;
        const long n = h->n, m = h->m, N = h->N;
        rp.A = q.A; rp.A_stride = N * n * n; rp.A_kstride = n * n;
        rp.B = q.B; rp.B_stride = N * n * m; rp.B_kstride = n * m;
        rp.c = q.c; rp.c_stride = N * n; rp.ebar = q.ebar; rp.ebar_stride = N * n;
        rp.qu = q.qadd; rp.qu_stride = N * m; rp.qu_scale = 0.5;
        rp.P = h->bP; rp.P_stride = q.sP;
        rp.x0 = nullptr; rp.flag = h->bFlag; rp.v_only = 1;
    }
    const size_t per = (size_t)rp.lds_per_wave * sizeof(double);
    int waves = RICCATI_WAVES;   // per-wave slices of LDS: as many waves per workgroup as fit
    while (waves > 1 && per * waves > 160 * 1024) --waves;
    const size_t lds = per * waves;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    int wgs = (h->batch + waves - 1) / waves;
    const int cap = h->num_cus * 2;
    if (wgs > cap) wgs = cap;
#define RICCATI_LAUNCH(NC_, MC_)                                                                               \
    do {                                                                                                       \
        const hipError_t e_ = ensure_dyn_lds(reinterpret_cast<const void*>(k_riccati_t<NC_, MC_>), lds);       \
        if (e_ != hipSuccess) return e_;                                                                       \
        hipLaunchKernelGGL((k_riccati_t<NC_, MC_>), dim3(wgs), dim3(64 * waves), lds, h->stream, rp);          \
    } while (0)
    if (h->n == 12 && h->m == 4 && !getenv("ALMPC_RICCATI_GENERIC")) RICCATI_LAUNCH(12, 4);
    else if (h->n == 4 && h->m == 2 && !getenv("ALMPC_RICCATI_GENERIC")) RICCATI_LAUNCH(4, 2);
    else if (h->n == 2 && h->m == 1 && !getenv("ALMPC_RICCATI_GENERIC")) RICCATI_LAUNCH(2, 1);
    else RICCATI_LAUNCH(0, 0);
#undef RICCATI_LAUNCH
    return hipGetLastError();
}

// ---- stage-wise dual active-set solve (k_sdual) ----------------------------------------------------------------------------
// instantiated (NT, MC): the smallest pair that covers (nt, m)
const int SD_SHAPES[][2] = {{2, 2}, {4, 2}, {6, 2}, {8, 4}, {12, 4}, {16, 4}, {16, 8}, {32, 16}, {48, 16}};
bool sdual_pick_shape(int nt, int m, int* NT, int* MC) {
    for (const auto& s : SD_SHAPES)
        if (nt <= s[0] && m <= s[1]) { *NT = s[0]; *MC = s[1]; return true; }
    return false;
}
constexpr int SD_WCAP1 = 32, SD_WCAP2 = 64, SD_WCAP3 = 96, SD_WCAP4 = 128;   // working-set capacity of the first launch / of the redos of the instances that outgrew it
bool sdual_shape_ok(int n, int m, int N, bool useS) {
    int NT = 0, MC = 0;
    if (!sdual_pick_shape(useS ? n + m : n, m, &NT, &MC)) return false;
    if ((N + 64 / (NT + MC)) / (64 / (NT + MC)) > 64) return false;   // (one bit per coordinate and lane in the working-set mask)
    return (size_t)sdual_lds_doubles(NT, MC, N, SD_WCAP2) * sizeof(double) <= 160 * 1024;
}

// Stage records of a SHARED model on the device (host Riccati, design time), state box / terminal equality / S of the design.
// Rm: the reference's branch rule applied (zeros when R[1,1] == 0); Sm null: no input-rate term.
hipError_t launch_sdual(almpc_handle* h, int filter, const double* guess, int max_iter, bool single_launch = false, int first_tier = 0);
hipError_t sdual_build_ghat(almpc_handle* h);

int sdual_setup_shared(almpc_handle* h, const hm::mat& Am, const hm::mat& Bm, const hm::mat& Qm, const hm::mat& Rm, const hm::mat* Sm,
                       const hm::mat& Pm, const double* xmin, const double* xmax, bool terminal_eq) {
    const int n = h->n, m = h->m, N = h->N;
    almpc_handle::Sd& sd = h->sd;
    sd.ready = false;
    sd.useS = Sm != nullptr;
    sd.nt = sd.useS ? n + m : n;
    if (!sdual_pick_shape(sd.nt, m, &sd.NT, &sd.MC) || !sdual_shape_ok(n, m, N, sd.useS))
        return fail(h, ALMPC_ERR_UNSUPPORTED, "stage-wise solve: n (+ m with an input-rate weight) <= 48, m <= 16 and (N + 1)(n + m) <= 4096");
    hm::mat rec;
    bool inv = false;
    if (!hm::stage_records(Am, Bm, Qm, Rm, Sm, Pm, n, m, N, sd.NT, sd.MC, rec, inv))
        return fail(h, ALMPC_ERR_NUMERIC, "stage-wise solve: R + B'PB is singular (Riccati recursion of the unconstrained problem)");
    const size_t stage = (size_t)sdual_rec_stage(sd.NT, sd.MC);
    const size_t cnt = inv ? stage : stage * N;
    if (sd.rec_cap < cnt) {
        if (sd.rec) { (void)hipFree(sd.rec); sd.rec = nullptr; }
        HIP_TRY(h, dalloc(&sd.rec, cnt));
        sd.rec_cap = cnt;
    }
    HIP_TRY(h, hipMemcpy(sd.rec, rec.data() + (inv ? stage * (N - 1) : 0), cnt * sizeof(double), hipMemcpyHostToDevice));
    sd.rec_stride = 0;
    sd.rec_kstride = inv ? 0 : (long)stage;
    sd.has_box = xmin != nullptr;
    if (sd.has_box) {
        if (!sd.xmin) HIP_TRY(h, dalloc(&sd.xmin, (size_t)n));
        if (!sd.xmax) HIP_TRY(h, dalloc(&sd.xmax, (size_t)n));
        HIP_TRY(h, hipMemcpy(sd.xmin, xmin, n * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(sd.xmax, xmax, n * sizeof(double), hipMemcpyHostToDevice));
    }
    sd.has_eq = terminal_eq;
    if (!sd.ovf) HIP_TRY(h, dalloc(&sd.ovf, (size_t)h->batch + 2));   // (+ the tiers' gate word: SdualParams::ovf_gate)
    if (!sd.wsave) HIP_TRY(h, dalloc(&sd.wsave, (size_t)h->batch * SDUAL_WSAVE));
    HIP_TRY(h, hipMemset(sd.ovf, 0, ((size_t)h->batch + 2) * sizeof(int32_t)));
    sd.S = Sm ? *Sm : hm::mat();
    sd.has_base = false; sd.base_stride = 0;
    sd.per_instance = false; sd.gain_N = 0; sd.sqp = false;
    sd.scr_ready = false;
    sd.ready = true;
    // cached responses (sdual_build_ghat): at design time for a structured handle, whose every step uses them; on a condensed handle,
    // where k_sdual is only the redo of what a step leaves undecided, at the first redo -- most such handles (the headline path) never
    // leave an instance undecided and should not pay TP^2 doubles and one sweep per coordinate at every design
    sd.ghat_ready = false;
    sd.ghat_wanted = !getenv("ALMPC_SDUAL_NO_GHAT");
    if (sd.ghat_wanted && (h->flags & ALMPC_FLAG_STRUCTURED)) {
        HIP_TRY(h, sdual_build_ghat(h));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return ALMPC_OK;
}

// Cached responses of a shared model: every coordinate that can be a row, one sweep each, once (k_sdual's build mode on
// min(TP, 2048) waves), TP x TP doubles.  Memory cap 256 MB per handle (TP <= 5792: e.g. the quadrotor with S up to N = 361); above it
// the table is not built and a working-set change costs two sweeps again (SdualParams::ghat == null).  Enqueued on the handle's stream.
hipError_t sdual_build_ghat(almpc_handle* h) {
    almpc_handle::Sd& sd = h->sd;
    sd.ghat_wanted = false;   // (one attempt per design)
    const size_t TP = (size_t)sdual_tp(sd.NT, sd.MC, h->N);
    if (TP * TP * sizeof(double) > ((size_t)256 << 20)) return hipSuccess;
    if (sd.ghat_cap < TP * TP) {
        if (sd.ghat) { (void)hipFree(sd.ghat); sd.ghat = nullptr; sd.ghat_cap = 0; }
        hipError_t e = dalloc(&sd.ghat, TP * TP);
        if (e != hipSuccess) return e;
        sd.ghat_cap = TP * TP;
    }
    hipError_t e = hipMemsetAsync(sd.ghat, 0, TP * TP * sizeof(double), h->stream);
    if (e != hipSuccess) return e;
    sd.ghat_building = true;
    e = launch_sdual(h, 0, nullptr, 0);
    sd.ghat_building = false;
    if (e != hipSuccess) return e;
    sd.ghat_ready = true;
    return hipSuccess;
}

// Linear cost terms of the stage-wise problem that depend on the references: only the input-rate term does (it is on u = v + u_ref,
// src/sub/design_mpc.jl:423-446): stage k >= 1 carries (w_k - v_k + dU)'S(w_k - v_k + dU), dU = u_ref[k-1] - u_ref[k]  ->
// +S dU on the coordinates of w_k = v_{k-1}, -S dU on those of v_k.  Zero for references that are constant over the horizon.
int sdual_update_base(almpc_handle* h, const double* uref, size_t cnt) {
    almpc_handle::Sd& sd = h->sd;
    sd.has_base = false; sd.base_stride = 0;
    sd.scr_ready = false;   // (the screen's interval table depends on the input reference)
    if (!sd.ready || !sd.useS) return ALMPC_OK;
    const int n = h->n, m = h->m, N = h->N, SP = sd.NT + sd.MC;
    const size_t TP = (size_t)sdual_tp(sd.NT, sd.MC, N), us = (size_t)h->nz;
    std::vector<double> base(cnt * TP, 0.0);
    bool any = false;
    for (size_t c = 0; c < cnt; ++c)
        for (int k = 1; k < N; ++k)
            for (int a = 0; a < m; ++a) {
                double sv = 0.0;
                for (int b = 0; b < m; ++b) sv += sd.S[(size_t)b * m + a] * (uref[c * us + (size_t)(k - 1) * m + b] - uref[c * us + (size_t)k * m + b]);
                if (sv != 0.0) any = true;
                base[c * TP + (size_t)k * SP + n + a] += sv;
                base[c * TP + (size_t)k * SP + sd.NT + a] -= sv;
            }
    if (!any) return ALMPC_OK;
    if (sd.base_cap < base.size()) {
        if (sd.base) { (void)hipFree(sd.base); sd.base = nullptr; }
        HIP_TRY(h, dalloc(&sd.base, base.size()));
        sd.base_cap = base.size();
    }
    HIP_TRY(h, hipMemcpy(sd.base, base.data(), base.size() * sizeof(double), hipMemcpyHostToDevice));
    sd.has_base = true;
    sd.base_stride = cnt > 1 ? (long)TP : 0;
    return ALMPC_OK;
}

// Weights and buffers of the stage-wise solve with a model PER INSTANCE (almpc_design_batched, the re-linearisation pipeline): the stage
// records come from k_sgains (launch_sgains) -- `invariant`: the terminal weight is every instance's own DARE solution and there is
// no input-rate weight, so one stage per instance suffices.  Rm: as given (the branch rule is applied here); Sm null: no rate term.
int sdual_setup_batched(almpc_handle* h, const hm::mat& Qm, const hm::mat& Rm, const hm::mat* Sm, bool invariant,
                        const double* xmin, const double* xmax, bool terminal_eq) {
    const int n = h->n, m = h->m, N = h->N;
    almpc_handle::Sd& sd = h->sd;
    sd.ready = false;
    sd.useS = Sm != nullptr;
    sd.nt = sd.useS ? n + m : n;
    if (!sdual_pick_shape(sd.nt, m, &sd.NT, &sd.MC) || !sdual_shape_ok(n, m, N, sd.useS))
        return fail(h, ALMPC_ERR_UNSUPPORTED, "stage-wise solve: n (+ m with an input-rate weight) <= 48, m <= 16 and (N + 1)(n + m) <= 4096");
    if ((size_t)sgains_lds_doubles(sd.nt, m) * sizeof(double) > 160 * 1024) return fail(h, ALMPC_ERR_UNSUPPORTED, "stage-wise solve: the gain recursion does not fit LDS");
    hm::mat Qs = Qm, Rs = Rm;
    if (Rm[0] == 0.0) std::fill(Rs.begin(), Rs.end(), 0.0);
    if (!sd.dQ) HIP_TRY(h, dalloc(&sd.dQ, (size_t)n * n));
    if (!sd.dR) HIP_TRY(h, dalloc(&sd.dR, (size_t)m * m));
    HIP_TRY(h, hipMemcpy(sd.dQ, Qs.data(), Qs.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(sd.dR, Rs.data(), Rs.size() * sizeof(double), hipMemcpyHostToDevice));
    if (Sm) {
        if (!sd.dS) HIP_TRY(h, dalloc(&sd.dS, (size_t)m * m));
        HIP_TRY(h, hipMemcpy(sd.dS, Sm->data(), Sm->size() * sizeof(double), hipMemcpyHostToDevice));
    }
    const size_t stage = (size_t)sdual_rec_stage(sd.NT, sd.MC);
    sd.gain_N = invariant ? 1 : N;
    const size_t cnt = (size_t)h->batch * sd.gain_N * stage;
    if (sd.rec_cap < cnt) {
        if (sd.rec) { (void)hipFree(sd.rec); sd.rec = nullptr; }
        HIP_TRY(h, dalloc(&sd.rec, cnt));
        sd.rec_cap = cnt;
    }
    sd.rec_stride = (long)(sd.gain_N * stage);
    sd.rec_kstride = invariant ? 0 : (long)stage;
    sd.per_instance = true; sd.sqp = false;
    sd.has_box = xmin != nullptr;
    if (sd.has_box) {
        if (!sd.xmin) HIP_TRY(h, dalloc(&sd.xmin, (size_t)n));
        if (!sd.xmax) HIP_TRY(h, dalloc(&sd.xmax, (size_t)n));
        HIP_TRY(h, hipMemcpy(sd.xmin, xmin, n * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(sd.xmax, xmax, n * sizeof(double), hipMemcpyHostToDevice));
    }
    sd.has_eq = terminal_eq;
    if (!sd.ovf) HIP_TRY(h, dalloc(&sd.ovf, (size_t)h->batch + 2));   // (+ the tiers' gate word: SdualParams::ovf_gate)
    HIP_TRY(h, hipMemset(sd.ovf, 0, ((size_t)h->batch + 2) * sizeof(int32_t)));
    if (!sd.wsave) HIP_TRY(h, dalloc(&sd.wsave, (size_t)h->batch * SDUAL_WSAVE));
    if (!sd.bad) HIP_TRY(h, dalloc(&sd.bad, (size_t)h->batch));
    HIP_TRY(h, hipMemset(sd.bad, 0, (size_t)h->batch * sizeof(int)));
    sd.S = Sm ? *Sm : hm::mat();
    sd.has_base = false; sd.base_stride = 0;
    sd.ready = true;
    return ALMPC_OK;
}

// k_sgains over the handle's per-instance models (bA, bB; terminal weights bP, shared or per instance): all instances
// (filter 0) or the ones the condensed step left unsolved (filter 1)
hipError_t launch_sgains(almpc_handle* h, int filter) {
    const almpc_handle::Sd& sd = h->sd;
    SgainsParams gp;
    std::memset(&gp, 0, sizeof(gp));
    const long n = h->n, m = h->m;
    gp.n = h->n; gp.nt = sd.nt; gp.m = h->m; gp.N = sd.gain_N; gp.batch = h->batch; gp.NT = sd.NT; gp.MC = sd.MC;
    gp.A = h->bA; gp.A_stride = n * n; gp.A_kstride = 0;
    gp.B = h->bB; gp.B_stride = n * m; gp.B_kstride = 0;
    gp.P = h->bP; gp.P_stride = h->rP_stride;   // (one matrix with stride 0, or one per instance)
    gp.Q = sd.dQ; gp.R = sd.dR; gp.S = sd.useS ? sd.dS : nullptr;
    gp.c = nullptr; gp.c_stride = 0;
    gp.filter = filter; gp.status = h->dStatus; gp.flag = nullptr;
    gp.rec = sd.rec; gp.rec_stride = sd.rec_stride;
    gp.pc = nullptr; gp.ct = nullptr; gp.pc_stride = 0;
    gp.bad = sd.bad;
    if (h->redo_gate_on) { gp.gate = h->dRedoGate; gp.gate_val = h->step_serial; }
    if (sd.sqp) {   // the QP of the current SQP iteration: stage models, defects, state errors and input gradient of the loop
        const almpc_handle::Sqp& q = h->sqp;
        const long N = h->N;
        gp.A = q.A; gp.A_stride = N * n * n; gp.A_kstride = n * n;
        gp.B = q.B; gp.B_stride = N * n * m; gp.B_kstride = n * m;
        gp.P = h->bP; gp.P_stride = q.sP;
        gp.c = q.c; gp.c_stride = N * n;
        gp.pc = sd.pc; gp.ct = sd.ct; gp.pc_stride = N * sd.NT;
        gp.ebar = q.ebar; gp.ebar_stride = N * n; gp.qadd = q.qadd; gp.qadd_stride = N * m; gp.qscale = 0.5;
        gp.base = sd.base; gp.base_stride = sd.base_stride;
        gp.flag = h->bFlag;
    }
    if (!gp.A || !gp.B || !gp.P || !gp.rec) return hipErrorInvalidValue;
    gp.lds_per_wave = sgains_lds_doubles(sd.nt, h->m);
    const size_t per = (size_t)gp.lds_per_wave * sizeof(double);
    int waves = SGAINS_WAVES;
    while (waves > 1 && per * waves > 160 * 1024) --waves;
    const size_t lds = per * waves;
    int wgs = (h->batch + waves - 1) / waves;
    const int cap = h->num_cus * 2;
    if (wgs > cap) wgs = cap;
    const hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(k_sgains), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_sgains, dim3(wgs), dim3(64 * waves), lds, h->stream, gp);
    return hipGetLastError();
}

template <int NT, int MC>
hipError_t launch_sdual_t(almpc_handle* h, SdualParams sp, int tier0, int tier1) {
    // tier 0: every (filtered) instance with room for SD_WCAP1 rows (several waves per workgroup); tier 1: the instances that outgrew it,
    // SD_WCAP2 rows; tier 2: SD_WCAP4 rows, two working-set positions per lane (mostly infeasible instances whose verdict needs that
    // many rows); tier 3: the same with Sinv in a global scratch, for shapes whose trajectories leave no room for it in LDS.
    // (tier0 = tier1 = 1: ONE launch with room for SD_WCAP2 rows -- the redo of the few instances a condensed step left unsolved)
    if (tier1 > tier0 && sp.wsave && !sp.build_ghat && !getenv("ALMPC_SDUAL_NO_SINV_HANDOVER")) {   // tiers hand each other the inverse of their working set
        if (!h->sd.sinv_save) {
            const hipError_t e_ = dalloc(&h->sd.sinv_save, (size_t)h->batch * sdual_sinv_doubles(SDUAL_SINV_SAVE));
            if (e_ != hipSuccess) return e_;
        }
        sp.sinv_save = h->sd.sinv_save;
    }
    // the later tiers return at once unless an earlier one of THIS call ran out of room (it stores the call's number in the gate word):
    // their scan for flagged instances alone is 16 dependent loads per wave, 9 us of an idle launch on 4096 instances
    if (sp.ovf && !sp.build_ghat) { sp.ovf_gate = sp.ovf + sp.batch; sp.ovf_gate_val = ++h->sd.tier_serial; }
    for (int tier = tier0; tier <= tier1; ++tier) {
        sp.wcap = tier == 0 ? SD_WCAP1 : (tier == 1 ? SD_WCAP2 : SD_WCAP4);
        sp.only_ovf = tier > tier0;
        bool glb = tier == 3;
        const bool fits128 = (size_t)sdual_lds_doubles(NT, MC, sp.N, SD_WCAP4, true) * sizeof(double) <= 160 * 1024;
        if (tier == 2 && !fits128) continue;   // (Sinv of 128 rows does not fit beside the trajectories: the global-scratch build)
        if (tier == 3 && fits128) continue;
        sp.lds_per_wave = sdual_lds_doubles(NT, MC, sp.N, sp.wcap, !glb);
        const size_t per = (size_t)sp.lds_per_wave * sizeof(double);
        int waves = SDUAL_WAVES;
        while (waves > 1 && per * waves > 160 * 1024) --waves;
        if (per * waves > 160 * 1024) return hipErrorInvalidValue;
        if (tier >= 1) waves = 1;   // few instances: spread them over the compute units
        const size_t lds = per * waves;
        int per_cu = (int)((160 * 1024) / lds);
        if (per_cu < 1) per_cu = 1;
        if (per_cu * waves > 8) per_cu = 8 / waves > 0 ? 8 / waves : 1;
        int wgs = (sp.batch + waves - 1) / waves;
        int cap = h->num_cus * per_cu;
        if (glb) cap = h->num_cus;
        if (wgs > cap) wgs = cap;
        hipError_t e;
        const size_t lds_gh = lds + (size_t)(NT + MC) * (sdual_rec_row(NT, MC) + 2) * sizeof(double);   // (+ the workgroup's copy of a stage-invariant record, rows padded)
        // cached responses: the build whose sweeps fetch their records stage by stage and whose column stream runs four times deeper (see k_sdual)
        const bool gh = sp.ghat != nullptr && lds_gh <= 160 * 1024 && !getenv("ALMPC_SDUAL_NO_GH");
        if (tier < 2) {
            const void* kf = gh ? reinterpret_cast<const void*>(k_sdual<NT, MC, 1, false, true>) : reinterpret_cast<const void*>(k_sdual<NT, MC, 1, false, false>);
            e = ensure_dyn_lds(kf, gh ? lds_gh : lds);
            if (e != hipSuccess) return e;
            if (gh) hipLaunchKernelGGL((k_sdual<NT, MC, 1, false, true>), dim3(wgs), dim3(64 * waves), lds_gh, h->stream, sp);
            else hipLaunchKernelGGL((k_sdual<NT, MC, 1, false, false>), dim3(wgs), dim3(64 * waves), lds, h->stream, sp);
        } else if (!glb) {
            const void* kf = gh ? reinterpret_cast<const void*>(k_sdual<NT, MC, 2, false, true>) : reinterpret_cast<const void*>(k_sdual<NT, MC, 2, false, false>);
            e = ensure_dyn_lds(kf, gh ? lds_gh : lds);
            if (e != hipSuccess) return e;
            if (gh) hipLaunchKernelGGL((k_sdual<NT, MC, 2, false, true>), dim3(wgs), dim3(64 * waves), lds_gh, h->stream, sp);
            else hipLaunchKernelGGL((k_sdual<NT, MC, 2, false, false>), dim3(wgs), dim3(64 * waves), lds, h->stream, sp);
        } else {
            sp.ghat = nullptr;   // (the global-scratch build has no cached-response variant: sweeps)
            const size_t need = (size_t)wgs * waves * sdual_sinv_doubles(SD_WCAP4);
            if (h->sd.sinv_cap < need) {
                if (h->sd.sinv_glb) { (void)hipFree(h->sd.sinv_glb); h->sd.sinv_glb = nullptr; h->sd.sinv_cap = 0; }
                e = dalloc(&h->sd.sinv_glb, need);
                if (e != hipSuccess) return e;
                h->sd.sinv_cap = need;
            }
            sp.sinv_glb = h->sd.sinv_glb;
            e = ensure_dyn_lds(reinterpret_cast<const void*>(k_sdual<NT, MC, 2, true>), lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_sdual<NT, MC, 2, true>), dim3(wgs), dim3(64 * waves), lds, h->stream, sp);
        }
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        sp.filter = 0;   // the later tiers select by the overflow flag alone (the first one has rewritten the statuses)
    }
    return hipSuccess;
}

// k_sdual over the batch (filter 0), over the instances whose status is not 0 (filter 1: redo after the condensed path), start from
// `guess` (inputs [batch][N][m]) when given
hipError_t launch_sdual(almpc_handle* h, int filter, const double* guess, int max_iter, bool single_launch, int first_tier) {
    if (!h->sd.ready) return hipErrorInvalidValue;
    if (h->sd.ghat_wanted && !h->sd.ghat_ready && !h->sd.ghat_building && !h->sd.per_instance && !h->sd.sqp) {   // first use on a condensed handle
        const hipError_t eb = sdual_build_ghat(h);
        if (eb != hipSuccess) return eb;
    }
    const almpc_handle::Sd& sd = h->sd;
    SdualParams sp;
    std::memset(&sp, 0, sizeof(sp));
    sp.n = h->n; sp.nt = sd.nt; sp.m = h->m; sp.N = h->N; sp.batch = h->batch;
    sp.rec = sd.rec; sp.rec_stride = sd.rec_stride; sp.rec_kstride = sd.rec_kstride;
    sp.base = sd.has_base ? sd.base : nullptr; sp.base_stride = sd.base_stride;
    sp.pc = nullptr; sp.ct = nullptr; sp.pc_stride = 0;
    sp.umin = h->dUmin; sp.umax = h->dUmax; sp.uref = h->dUref; sp.uref_stride = h->uref_stride;
    sp.xmin = sd.has_box ? sd.xmin : nullptr; sp.xmax = sd.has_box ? sd.xmax : nullptr;
    sp.xbref = h->dXref; sp.xbref_stride = h->xref_stride;
    sp.eqt = sd.has_eq ? h->dXref + (size_t)h->N * h->n : nullptr; sp.eqt_stride = h->xref_stride;   // x_N = x_ref_N
    sp.x0 = h->dX0; sp.x0_stride = h->n; sp.xref = h->dXref; sp.xref_stride = h->xref_stride;
    if (h->redo_x0_from_results) { sp.x0 = h->dX; sp.x0_stride = (long)h->n * (h->N + 1); }   // (lazy redo: see resolve_lazy_redo)
    if (h->redo_gate_on) { sp.gate = h->dRedoGate; sp.gate_val = h->step_serial; }
    sp.uguess = guess; sp.filter = filter; sp.flag = nullptr; sp.v_only = 0;
    if (filter == 1 && sd.start_ws && sd.start_ws_fresh && !sd.sqp) sp.start_ws = sd.start_ws;   // (the redo behind a state-row finish)
    sp.x = h->dX; sp.ex = h->dEx; sp.u = h->dU; sp.eu = h->dEu; sp.status = h->dStatus; sp.piters = h->dPiters;
    if (sd.sqp) {   // dx_0 = 0, variable v = u - ubar (the handle's per-instance references ARE the iterate xbar, ubar), cost terms per instance
        const almpc_handle::Sqp& q = h->sqp;
        sp.base = sd.base; sp.base_stride = sd.base_stride;
        sp.pc = sd.pc; sp.ct = sd.ct; sp.pc_stride = (long)h->N * sd.NT;
        sp.eqt = sd.has_eq ? q.xref + (size_t)h->N * h->n : nullptr; sp.eqt_stride = 0;
        sp.x0 = nullptr; sp.xref = nullptr; sp.xref_stride = 0;
        sp.flag = h->bFlag; sp.v_only = 1;
    }
    sp.ovf = sd.ovf; sp.only_ovf = 0; sp.wsave = sd.wsave;
    sp.gbad = sd.per_instance ? sd.bad : nullptr;
    sp.ghat = (sd.ghat_ready && !sd.per_instance && !sd.sqp && sd.rec_stride == 0) ? sd.ghat : nullptr;
    if (sd.ghat_building) {   // design time: the waves walk the coordinates (no instance data is read)
        const int TP = sdual_tp(sd.NT, sd.MC, h->N);
        sp.build_ghat = 1; sp.ghat_out = sd.ghat; sp.ghat = nullptr;
        sp.batch = TP < 2048 ? TP : 2048;
        sp.uref_stride = 0; sp.xbref = nullptr; sp.xbref_stride = 0; sp.x0 = nullptr; sp.xref = nullptr; sp.xref_stride = 0;
        sp.eqt = sd.has_eq ? sd.rec : nullptr; sp.eqt_stride = 0;   // (has_eq decides which coordinates can be rows; the target values are not used)
        sp.umin = sd.rec; sp.umax = sd.rec; sp.uref = sd.rec;       // (read into the bound tables, not used: the handle's own arrays may not exist yet)
        sp.x = sp.ex = sp.u = sp.eu = nullptr; sp.status = nullptr; sp.piters = nullptr;
        sp.base = nullptr; sp.pc = nullptr; sp.ct = nullptr; sp.uguess = nullptr; sp.filter = 0; sp.flag = nullptr;
        sp.ovf = nullptr; sp.wsave = nullptr; sp.gbad = nullptr;
        single_launch = false; first_tier = 0;
    }
    sp.rows_state = (sd.has_box || sd.has_eq) ? 1 : 0;
    // state box of a shared model with shared references, every instance solved from scratch: the reachability screen first (one
    // table kernel per design / reference change, one small launch per solve) -- instances it certifies infeasible never reach a sweep
    if (sd.has_box && !sd.ghat_building && !sd.per_instance && !sd.sqp && filter == 0 && first_tier == 0 && sp.x0 && h->uref_stride == 0 &&
        h->xref_stride == 0 && h->dA && h->dB && !getenv("ALMPC_SDUAL_NO_SCREEN")) {
        almpc_handle::Sd& sdw = h->sd;
        const size_t n_ = (size_t)h->n, m_ = (size_t)h->m, N_ = (size_t)h->N;
        hipError_t e;
        if (sdw.scr_cap < N_) {
            for (double** q : {&sdw.scr_phi, &sdw.scr_g, &sdw.scr_rm, &sdw.scr_rp}) if (*q) { (void)hipFree(*q); *q = nullptr; }
            if ((e = dalloc(&sdw.scr_phi, N_ * n_ * n_)) != hipSuccess || (e = dalloc(&sdw.scr_g, N_ * n_ * m_)) != hipSuccess ||
                (e = dalloc(&sdw.scr_rm, N_ * n_)) != hipSuccess || (e = dalloc(&sdw.scr_rp, N_ * n_)) != hipSuccess) return e;
            sdw.scr_cap = N_; sdw.scr_ready = false;
        }
        if (!sdw.scr_verdict && (e = dalloc(&sdw.scr_verdict, (size_t)h->batch)) != hipSuccess) return e;
        ScreenParams cp;
        cp.n = h->n; cp.m = h->m; cp.N = h->N; cp.batch = h->batch;
        cp.A = h->dA; cp.B = h->dB; cp.umin = h->dUmin; cp.umax = h->dUmax; cp.uref = h->dUref;
        cp.xmin = sd.xmin; cp.xmax = sd.xmax; cp.xref = h->dXref; cp.x0 = sp.x0; cp.x0_stride = sp.x0_stride;
        cp.phi = sdw.scr_phi; cp.gtab = sdw.scr_g; cp.rm = sdw.scr_rm; cp.rp = sdw.scr_rp; cp.verdict = sdw.scr_verdict;
        if (!sdw.scr_ready) {
            hipLaunchKernelGGL(k_screen_tables, dim3(1), dim3(256), 0, h->stream, cp);
            if ((e = hipGetLastError()) != hipSuccess) return e;
            sdw.scr_ready = true;
        }
        if ((e = hipMemsetAsync(sdw.scr_verdict, 0, (size_t)h->batch * sizeof(int32_t), h->stream)) != hipSuccess) return e;
        const dim3 sg((unsigned)((h->batch + 63) / 64), (unsigned)h->N);
#define SCR_CASE(NT_) if (h->n <= NT_) hipLaunchKernelGGL((k_state_box_screen<NT_>), sg, dim3(64), 0, h->stream, cp)
        SCR_CASE(4); else SCR_CASE(8); else SCR_CASE(12); else SCR_CASE(16); else SCR_CASE(32); else SCR_CASE(48); else return hipErrorInvalidValue;
#undef SCR_CASE
        if ((e = hipGetLastError()) != hipSuccess) return e;
        sp.screen = sdw.scr_verdict;
    }
    const int rows = h->N * h->m + ((sd.has_box || sd.has_eq) ? h->N * h->n : 0);
    sp.max_iter = max_iter > 0 ? max_iter : 20 * rows + 50;
    sp.tol = 1e-9;
    // single launch (the redo behind a condensed step: few instances, occupancy does not matter): the 128-row build when its Sinv fits
    // LDS beside the trajectories, else the 64-row one.  first_tier 1: starts that are known to hold many rows
    int tier0 = first_tier, tier1 = 3;
    if (sd.ghat_building) tier1 = 0;
    if (single_launch) {
        int NT_ = sd.NT, MC_ = sd.MC;
        const bool fits128 = (size_t)sdual_lds_doubles(NT_, MC_, h->N, SD_WCAP4, true) * sizeof(double) <= 160 * 1024;
        // (round 5: the 64-row build first -- one working-set position per lane: its Sinv products, borderings and column streams are
        // cheaper per change than the 128-row build's -- and the 128-row build only for what outgrows it, which costs little since the
        // tiers hand over their inverse; ALMPC_SDUAL_REDO_128=1: the one 128-row launch of round 4)
        tier0 = (fits128 && (getenv("ALMPC_SDUAL_REDO_128") || (h->redo_gate_on && !h->redo_predicted))) ? 2 : 1;   // (a gated redo that is not expected to have work: one launch)
        tier1 = fits128 ? 2 : 1;
    }
    // the redo's start (the finish's working set + the terminal-equality rows) with its inverse, built in registers from the cached
    // responses before the solve: k_sdual_start (csrc/almpc_sdual.hip.h)
    if (sp.start_ws && sp.ghat && tier0 >= 1 && !getenv("ALMPC_SDUAL_NO_START_BUILD")) {
        almpc_handle::Sd& sdw = h->sd;
        if (!sdw.start_inv) {
            const hipError_t e_ = dalloc(&sdw.start_inv, (size_t)h->batch * sdual_sinv_doubles(SDUAL_SINV_SAVE));
            if (e_ != hipSuccess) return e_;
        }
        SdualStartParams tp;
        std::memset(&tp, 0, sizeof(tp));
        tp.batch = h->batch; tp.n = h->n; tp.N = h->N; tp.SP = sd.NT + sd.MC; tp.TP = sdual_tp(sd.NT, sd.MC, h->N);
        // (the list is built for the LARGEST tier of this call: the first tier installs it if it leaves room to work, else hands it on)
        tp.wcap = (tier0 == 1 && tier1 < 2) ? SD_WCAP2 : SD_WCAP4; tp.has_eq = sd.has_eq ? 1 : 0;
        tp.status = h->dStatus; tp.gate = sp.gate; tp.gate_val = sp.gate_val;
        tp.ghat = sp.ghat; tp.start_ws = sdw.start_ws; tp.start_inv = sdw.start_inv;
        // (one workgroup per instance at a time; a workgroup looks at up to 64 SDUAL_START_WAVES instances)
        int wgs = h->batch < h->num_cus * 4 ? h->batch : h->num_cus * 4;
        const int wmin = (h->batch + 64 * SDUAL_START_WAVES - 1) / (64 * SDUAL_START_WAVES);
        if (wgs < wmin) wgs = wmin;
        hipLaunchKernelGGL(k_sdual_start, dim3(wgs), dim3(64 * SDUAL_START_WAVES), 0, h->stream, tp);
        const hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) return e_;
        sp.start_inv = sdw.start_inv;
    }
#define SD_CASE(NT_, MC_) if (sd.NT == NT_ && sd.MC == MC_) return launch_sdual_t<NT_, MC_>(h, sp, tier0, tier1)
    SD_CASE(2, 2); SD_CASE(4, 2); SD_CASE(6, 2); SD_CASE(8, 4); SD_CASE(12, 4); SD_CASE(16, 4); SD_CASE(16, 8); SD_CASE(32, 16); SD_CASE(48, 16);
#undef SD_CASE
    return hipErrorInvalidValue;
}

// Wait for the handle's stream: poll for a while before the blocking wait, whose wake-up (interrupt + scheduler) costs 0.1 - 15 ms on
// this pool -- more than the kernels waited for.
hipError_t stream_wait_polling(almpc_handle* h) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(h->stream);
        if (q != hipErrorNotReady) return q;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
    }
    return hipStreamSynchronize(h->stream);
}

// Lazy redo (see almpc_handle::hUnsolved): called where the host is about to look at results and the stream is idle.  If the finish of
// a step since the last look left instances unsolved, the stage-wise solvers redo them now (from the step's own result).
int resolve_lazy_redo(almpc_handle* h) {
    if (!h->lazy_pending) return ALMPC_OK;
    h->lazy_pending = false;
    if (!h->hUnsolved) return ALMPC_OK;
    const int cur = *reinterpret_cast<volatile int*>(h->hUnsolved);
    h->redo_expected = cur != h->unsolved_seen;
    if (cur == h->unsolved_seen) return ALMPC_OK;
    h->unsolved_seen = cur;
    h->redo_x0_from_results = true;
    struct Reset { almpc_handle* h; ~Reset() { h->redo_x0_from_results = false; } } reset_{h};
    if (h->sd.ready) {
        if (h->sd.per_instance) HIP_TRY(h, launch_sgains(h, 1));   // (the last step's models are still in the model slots)
        HIP_TRY(h, launch_sdual(h, 1, h->dU, 0, true));
    }
    if (h->mc == 0 && !h->useS && h->rKst) HIP_TRY(h, launch_riccati(h, 1, h->dU, 0));
    HIP_TRY(h, stream_wait_polling(h));   // (a redo is 0.1 - 0.4 ms of kernels: the blocking wait's wake-up alone was seen to double that)
    return ALMPC_OK;
}

// Gated redo behind the last enqueued step (no host look, no synchronisation): see almpc_handle::dRedoGate.  Called where the step's
// results are about to be consumed without a synchronous call; does nothing unless a lazily deferred redo is pending.
int enqueue_gated_redo(almpc_handle* h) {
    if (!h->lazy_pending || !h->dRedoGate || getenv("ALMPC_NO_GATED_REDO")) return ALMPC_OK;
    h->redo_x0_from_results = true; h->redo_gate_on = true;
    struct Reset { almpc_handle* h; ~Reset() { h->redo_x0_from_results = false; h->redo_gate_on = false; } } reset_{h};
    // As few launches as possible: an empty gated launch still costs 3 - 4 us of stream time on a 60 us step.  An input box alone
    // (no state rows, no S): the primal Riccati active set by itself -- it is the solver that needs no certificate from another one
    // (slower per instance than the dual method, but what it gets here is rare) --, ONE launch.  Otherwise the dual method's
    // 128-row build in one launch (+ the stage records of per-instance models).
    if (h->mc == 0 && !h->useS && h->rKst) HIP_TRY(h, launch_riccati(h, 1, h->dU, 0));
    else if (h->sd.ready) {
        if (h->sd.per_instance) HIP_TRY(h, launch_sgains(h, 1));
        HIP_TRY(h, launch_sdual(h, 1, h->dU, 0, true));
    }
    h->lazy_pending = false;   // (this step is settled on the stream; a later synchronous look has nothing left to do for it)
    return ALMPC_OK;
}

// Synchronous look: wait for the handle's stream, then settle the last step's deferred redo.  A closed loop that leaves instances
// undecided tends to leave them step after step (the same plants sit at the edge of feasibility), and a redo enqueued AFTER the host has
// seen the count pays a wake-up and a launch latency with the GPU idle: when the previous look found undecided instances, the redo goes
// on the stream GATED behind the step before the wait -- it runs without the host in between if the step left anything, and costs two
// idle launches if it did not.  The look itself then only reads the count, to know what to expect of the next step.
int wait_and_settle(almpc_handle* h, bool blocking_only = false) {
    bool predicted = false;
    if (h->lazy_pending && h->redo_expected && h->dRedoGate && h->hUnsolved && !getenv("ALMPC_NO_PREDICTED_REDO")) {
        h->redo_predicted = true;
        const int rc = enqueue_gated_redo(h);
        h->redo_predicted = false;
        if (rc != ALMPC_OK) return rc;
        predicted = !h->lazy_pending;
    }
    if (blocking_only) HIP_TRY(h, hipStreamSynchronize(h->stream));
    else HIP_TRY(h, stream_wait_polling(h));
    if (predicted) {
        const int cur = *reinterpret_cast<volatile int*>(h->hUnsolved);
        h->redo_expected = cur != h->unsolved_seen;
        h->unsolved_seen = cur;
        return ALMPC_OK;
    }
    return resolve_lazy_redo(h);
}

// A new design voids a redo that was deferred for a step of the previous one (its models, references and results are about to go)
void drop_lazy_redo(almpc_handle* h) {
    h->lazy_pending = false;
    h->redo_expected = false;
    if (h->hUnsolved) h->unsolved_seen = *reinterpret_cast<volatile int*>(h->hUnsolved);
}

// Table of the state rows' s0 (PolishGenParams::s0_basis) for a shared design: after the design and after every change of the shared
// references (v0S).  Leaves s0_basis_ok false where the table does not apply (the finish then rolls v0 out).
int build_s0_basis(almpc_handle* h) {
    h->s0_basis_ok = false;
    if (h->mc <= 0 || h->batched || h->ltv || h->structured || !h->dVsPlain || !h->dRowTraj || getenv("ALMPC_NO_S0_BASIS")) return ALMPC_OK;
    int roll_g = 1;
    while (2 * roll_g * h->n <= 64) roll_g *= 2;
    const int roll_C = h->n + h->m;
    int roll_cpl = (roll_C + roll_g - 1) / roll_g;
    if (!((size_t)(h->N + 1) * roll_C <= 32 * 32 && roll_cpl <= 8)) return ALMPC_OK;
    roll_cpl = roll_cpl <= 1 ? 1 : (roll_cpl <= 2 ? 2 : (roll_cpl <= 4 ? 4 : 8));
    if (!h->dS0Basis) HIP_TRY(h, dalloc(&h->dS0Basis, (size_t)(h->n + 1) * h->Rs));
    S0BasisParams bp;
    bp.n = h->n; bp.m = h->m; bp.N = h->N; bp.nz = h->nz; bp.nzs = h->nzs; bp.R = h->R; bp.Rs = h->Rs; bp.roll_g = roll_g; bp.roll_cpl = roll_cpl;
    bp.A = h->dA; bp.B = h->dB; bp.Vs = h->dVsPlain; bp.v0S = (h->dV0S && h->fS_stride == 0) ? h->dV0S : nullptr; bp.dvec = h->dD;
    bp.row_traj = h->dRowTraj; bp.out = h->dS0Basis;
    hipLaunchKernelGGL(k_s0_basis, dim3((unsigned)(h->n + 1)), dim3(64), (size_t)(h->N + 1) * roll_C * sizeof(double), h->stream, bp);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->s0_basis_ok = true;
    return ALMPC_OK;
}

}  // namespace

extern "C" {

void almpc_default_opts(almpc_opts* o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->rho = 0.1;
    o->sigma = 1e-6;
    o->alpha = 1.6;
    o->eps_abs = 1e-3;
    o->eps_rel = 1e-3;
    o->max_iter = 25;
    o->check_every = 25;
    o->polish = 1;
    o->polish_max_iter = 0;
    o->warm_start = 0;
}

const char* almpc_last_error(const almpc_handle* h) { return h ? h->err.c_str() : "null handle"; }

int almpc_create(almpc_handle** out, int n, int m, int N, int batch, int device_id, uint32_t flags) {
    if (!out) return ALMPC_ERR_INVALID;
    *out = nullptr;
    if (n < 1 || m < 1 || N < 1 || batch < 1) return ALMPC_ERR_INVALID;
    const bool structured = (flags & ALMPC_FLAG_STRUCTURED) != 0;
    if (structured) {
        if (n > 32 || m > 16 || (long)m * N > 1024 || riccati_lds_doubles(n, m, N) * sizeof(double) > 160 * 1024) return ALMPC_ERR_UNSUPPORTED;
    } else if (n > 64 || (long)m * N > 128) return ALMPC_ERR_UNSUPPORTED;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) return ALMPC_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= count) return ALMPC_ERR_NO_DEVICE;
    almpc_handle* h = new almpc_handle();
    h->n = n; h->m = m; h->N = N; h->batch = batch; h->device = device_id; h->flags = flags;
    h->structured = structured;
    h->nz = m * N;
    h->nrb = (h->nz + 15) / 16;
    h->nzs = 16 * h->nrb;
    h->ks = pick_ks(h->nz, h->nrb);
    h->ksf = (n + 3) / 4;
    auto bail = [&](int code, const std::string& msg) {
        std::fprintf(stderr, "almpc_create: %s\n", msg.c_str());
        free_all(h);
        delete h;
        return code;
    };
#define TRY(call)                                                                  \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) return bail(ALMPC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
    TRY(hipSetDevice(device_id));
    TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    {
        int cus = 0;
        TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id));
        if (cus > 0) h->num_cus = cus;
        const char* e = getenv("ALMPC_POLISH_NO_GLDS");
        h->polish_no_glds = (e && e[0] == '1') ? 1 : 0;
        const char* e2 = getenv("ALMPC_NO_FUSED_STEP");
        h->fuse_step = (e2 && e2[0] == '1') ? 0 : 1;
    }
    const size_t fr = (size_t)h->nrb * h->ks * 64, b = (size_t)batch;
    if (structured) {   // no condensed matrices: models, references, inputs / outputs and the gain scratch only
        TRY(dalloc(&h->dUmin, (size_t)m)); TRY(dalloc(&h->dUmax, (size_t)m));
        TRY(dalloc(&h->dA, (size_t)n * n)); TRY(dalloc(&h->dB, (size_t)n * m));
        TRY(dalloc(&h->dX0, b * n));
        TRY(dalloc(&h->dX, b * n * (N + 1))); TRY(dalloc(&h->dEx, b * n * (N + 1)));
        TRY(dalloc(&h->dU, b * h->nz)); TRY(dalloc(&h->dEu, b * h->nz));
        TRY(dalloc(&h->dStatus, b)); TRY(dalloc(&h->dIters, b)); TRY(dalloc(&h->dPiters, b));
        TRY(hipMemset(h->dIters, 0, b * sizeof(int32_t)));
        TRY(hipMemset(h->dX0, 0, b * n * sizeof(double)));
        *out = h;
        return ALMPC_OK;
    }
    TRY(dalloc(&h->dMinvFrag, fr)); TRY(dalloc(&h->dVFrag, (size_t)h->nrb * h->ksf * 64)); TRY(dalloc(&h->dHFrag, fr));
    TRY(dalloc(&h->dFFrag, (size_t)h->nrb * h->ksf * 64));
    TRY(dalloc(&h->dG, (size_t)h->nz * h->nzs));
    TRY(dalloc(&h->dD, (size_t)h->nzs)); TRY(dalloc(&h->dRho, (size_t)h->nzs)); TRY(dalloc(&h->dUmin, (size_t)m)); TRY(dalloc(&h->dUmax, (size_t)m));
    TRY(dalloc(&h->dA, (size_t)n * n)); TRY(dalloc(&h->dB, (size_t)n * m));
    TRY(dalloc(&h->dX0, b * n));
    TRY(dalloc(&h->dXs, b * h->nzs)); TRY(dalloc(&h->dZs, b * h->nzs)); TRY(dalloc(&h->dYs, b * h->nzs));
    TRY(dalloc(&h->dV0, b * h->nzs)); TRY(dalloc(&h->dW, b * h->nzs));
    TRY(dalloc(&h->dX, b * n * (N + 1))); TRY(dalloc(&h->dEx, b * n * (N + 1)));
    TRY(dalloc(&h->dU, b * h->nz)); TRY(dalloc(&h->dEu, b * h->nz));
    TRY(dalloc(&h->dSglobal, b * POLISH_GLB_PER_INST));
    TRY(dalloc(&h->dPerm, ((b + 15) / 16) * 16));
    TRY(hipMemset(h->dPerm, 0xFF, ((b + 15) / 16) * 16 * sizeof(int32_t)));
    TRY(dalloc(&h->dStatus, b)); TRY(dalloc(&h->dIters, b)); TRY(dalloc(&h->dPiters, b));
    TRY(dalloc(&h->dYflags, b * h->nrb));
    TRY(hipMemset(h->dXs, 0, b * h->nzs * sizeof(double)));
    TRY(hipMemset(h->dZs, 0, b * h->nzs * sizeof(double)));
    TRY(hipMemset(h->dYs, 0, b * h->nzs * sizeof(double)));
    TRY(hipMemset(h->dX0, 0, b * n * sizeof(double)));
#undef TRY
    *out = h;
    return ALMPC_OK;
}

void almpc_destroy(almpc_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    free_all(h);
    delete h;
}

int almpc_set_rho_profile(almpc_handle* h, int mode) {
    if (!h) return ALMPC_ERR_INVALID;
    if (mode != 0 && mode != 1) return fail(h, ALMPC_ERR_INVALID, "rho profile: 0 (scalar) or 1 (stiffness)");
    h->rho_mode = mode;
    h->designed = false;  // takes effect at the next design
    return ALMPC_OK;
}

int almpc_set_step_fusion(almpc_handle* h, int on) {
    if (!h) return ALMPC_ERR_INVALID;
    h->fuse_step = on ? 1 : 0;
    return ALMPC_OK;
}

int almpc_set_terminal_equality(almpc_handle* h, int on) {
    if (!h) return ALMPC_ERR_INVALID;
    h->terminal_eq = on ? 1 : 0;
    h->designed = false;  // takes effect at the next design
    return ALMPC_OK;
}

int almpc_set_state_box(almpc_handle* h, const double* xmin, const double* xmax) {
    if (!h) return ALMPC_ERR_INVALID;
    if ((xmin == nullptr) != (xmax == nullptr)) return fail(h, ALMPC_ERR_INVALID, "set_state_box: give both xmin and xmax or neither");
    if (xmin) {
        for (int i = 0; i < h->n; ++i)
            if (!(xmin[i] <= xmax[i])) return fail(h, ALMPC_ERR_INVALID, "set_state_box: xmin > xmax");
        h->boxmin.assign(xmin, xmin + h->n); h->boxmax.assign(xmax, xmax + h->n);
    } else { h->boxmin.clear(); h->boxmax.clear(); }
    h->designed = false;  // takes effect at the next per-instance / time-varying / SQP design
    h->sqp.ready = h->sqp.started = false;
    h->relin.ready = false;
    return ALMPC_OK;
}

int almpc_set_structured_fallback(almpc_handle* h, int on) {
    if (!h) return ALMPC_ERR_INVALID;
    if (on && !riccati_shape_ok(h) && !sdual_shape_ok(h->n, h->m, h->N, false))
        return fail(h, ALMPC_ERR_UNSUPPORTED, "structured fallback: the shape is outside both stage-wise solvers");
    h->fallback = on ? 1 : 0;
    h->designed = false;  // takes effect at the next design (which prepares the weights on the device)
    return ALMPC_OK;
}

int almpc_set_start_from(almpc_handle* h, almpc_handle* src) {
    if (!h || !src) return ALMPC_ERR_INVALID;
    if (!h->structured) return fail(h, ALMPC_ERR_UNSUPPORTED, "set_start_from: the handle must be a structured one (ALMPC_FLAG_STRUCTURED)");
    if (!h->designed || !src->designed) return fail(h, ALMPC_ERR_NOT_DESIGNED, "set_start_from before design (of either handle)");
    if (src->n != h->n || src->m != h->m || src->batch != h->batch || src->N > h->N || src->device != h->device)
        return fail(h, ALMPC_ERR_INVALID, "set_start_from: the source must have the same n, m, batch and device and a horizon <= this handle's");
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->rGuess) HIP_TRY(h, dalloc(&h->rGuess, (size_t)h->batch * h->nz));
    if (!h->ev_guess) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_guess, hipEventDisableTiming));
    // the source's last step must be done before its inputs are read on this handle's stream (no host wait)
    HIP_TRY(h, hipEventRecord(h->ev_guess, src->stream));
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_guess, 0));
    const long cnt = (long)h->batch * h->nz;
    hipLaunchKernelGGL(k_guess_from_inputs, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, h->batch, h->m, src->N, h->N, 0,
                       (const double*)src->dU, (const double*)h->dUref, h->uref_stride, h->rGuess);
    HIP_TRY(h, hipGetLastError());
    // ... and src's NEXT step must not overwrite its inputs before that kernel has read them: src's stream waits for it
    if (!h->ev_guess_done) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_guess_done, hipEventDisableTiming));
    HIP_TRY(h, hipEventRecord(h->ev_guess_done, h->stream));
    HIP_TRY(h, hipStreamWaitEvent(src->stream, h->ev_guess_done, 0));
    h->guess_ready = true;
    return ALMPC_OK;
}

int almpc_design_shared(almpc_handle* h, const double* A, const double* B, const double* Q, const double* R,
                        const double* S, const double* P, const double* umin, const double* umax,
                        const double* xmin, const double* xmax, double rho, double sigma) {
    if (!h) return ALMPC_ERR_INVALID;
    drop_lazy_redo(h);
    if (!A || !B || !Q || !R || !umin || !umax) return fail(h, ALMPC_ERR_INVALID, "design: null matrix pointer");
    if (h->structured) {   // ALMPC_FLAG_STRUCTURED: no condensed matrices at all; rho / sigma are not used
        const int n = h->n, m = h->m, N = h->N;
        if ((xmin == nullptr) != (xmax == nullptr)) return fail(h, ALMPC_ERR_INVALID, "design: give both xmin and xmax or neither");
        for (int i = 0; i < m; ++i)
            if (!(umin[i] <= umax[i])) return fail(h, ALMPC_ERR_INVALID, "design: umin > umax");
        if (xmin)
            for (int i = 0; i < n; ++i)
                if (!(xmin[i] <= xmax[i])) return fail(h, ALMPC_ERR_INVALID, "design: xmin > xmax");
        h->designed = false;
        h->r_has_step = false; h->guess_ready = false;
        h->sqp.ready = h->sqp.started = false;   // (an SQP loop set up on this handle is gone with its per-instance reference buffers)
        HIP_TRY(h, hipSetDevice(h->device));
        hm::mat Am(A, A + (size_t)n * n), Bm(B, B + (size_t)n * m), Qm(Q, Q + (size_t)n * n), Rm(R, R + (size_t)m * m), Pm;
        if (P) Pm.assign(P, P + (size_t)n * n);
        else if (!hm::dare(Am, Bm, Qm, Rm, n, m, Pm)) return fail(h, ALMPC_ERR_NUMERIC, "design: DARE did not converge");
        h->P = Pm; h->H.clear(); h->F.clear(); h->d.clear();
        // the reference's branch rules (src/sub/design_mpc.jl:423-466): R and S enter only with a non-zero [1,1] element, S only with R
        const bool useS = R[0] != 0.0 && S && S[0] != 0.0;
        hm::mat Rb = Rm, Sm;
        if (Rm[0] == 0.0) std::fill(Rb.begin(), Rb.end(), 0.0);
        if (useS) Sm.assign(S, S + (size_t)m * m);
        h->sd.ready = false;
        if (sdual_shape_ok(n, m, N, useS) && !getenv("ALMPC_STRUCTURED_PRIMAL")) {
            // the stage-wise dual active set (k_sdual): input box, state box, terminal equality, input-rate weight
            const int rc_ = sdual_setup_shared(h, Am, Bm, Qm, Rb, useS ? &Sm : nullptr, Pm, xmin, xmax, h->terminal_eq != 0);
            if (rc_ != ALMPC_OK) return rc_;
        } else if (xmin || h->terminal_eq || useS)
            return fail(h, ALMPC_ERR_UNSUPPORTED, "structured solve: state rows / input-rate weight need the stage-wise dual solve (n + m <= 48, (N + 1)(n + m) <= 4096)");
        // the primal Riccati active set (k_riccati, input box only): the solver of shapes k_sdual does not cover and the safety net for
        // instances the dual method leaves without a certificate (a saturated open-loop unstable plant: Ghat_WW numerically singular)
        if (riccati_shape_ok(h)) { const int rc_ = riccati_weights(h, Qm, Rm, Pm.data()); if (rc_ != ALMPC_OK) return rc_; }
        else if (!h->sd.ready) return fail(h, ALMPC_ERR_UNSUPPORTED, "structured solve: shape outside both stage-wise solvers");
        HIP_TRY(h, hipMemcpy(h->dUmin, umin, m * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->dUmax, umax, m * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->dA, A, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->dB, B, (size_t)n * m * sizeof(double), hipMemcpyHostToDevice));
        h->batched = false; h->ltv = false; h->r_batched_P = false; h->rP_stride = 0;
        h->hS.assign((size_t)m * m, 0.0); h->useS = useS ? 1 : 0;
        if (useS) h->hS = Sm;
        h->has_box = xmin ? 1 : 0;
        h->designed = true;
        std::vector<double> xr((size_t)n * (N + 1), 0.0), ur((size_t)h->nz, 0.0);
        const int rc_ref = almpc_set_reference(h, xr.data(), ur.data(), 0);
        if (rc_ref != ALMPC_OK) h->designed = false;
        return rc_ref;
    }
    if ((xmin == nullptr) != (xmax == nullptr)) return fail(h, ALMPC_ERR_INVALID, "design: give both xmin and xmax or neither");
    if (!(rho > 0.0) || !(sigma >= 0.0)) return fail(h, ALMPC_ERR_INVALID, "design: rho must be > 0 and sigma >= 0");
    // from here on the handle's previous design is being overwritten: it counts as designed again only after the last step below
    h->designed = false;
    h->sqp.ready = h->sqp.started = false;
    h->relin.ready = false;
    HIP_TRY(h, hipSetDevice(h->device));
    const int n = h->n, m = h->m, N = h->N, nz = h->nz;
    for (int i = 0; i < m; ++i)
        if (!(umin[i] <= umax[i])) return fail(h, ALMPC_ERR_INVALID, "design: umin > umax");
    hm::mat Am(A, A + (size_t)n * n), Bm(B, B + (size_t)n * m), Qm(Q, Q + (size_t)n * n), Rm(R, R + (size_t)m * m);
    hm::mat Sm = S ? hm::mat(S, S + (size_t)m * m) : hm::mat((size_t)m * m, 0.0);
    hm::mat Pm;
    if (P) Pm.assign(P, P + (size_t)n * n);
    else if (!hm::dare(Am, Bm, Qm, Rm, n, m, Pm)) return fail(h, ALMPC_ERR_NUMERIC, "design: DARE did not converge");
    h->P = Pm;
    for (int j = 0; j < m; ++j)  // S enters the cost as a quadratic form: only its symmetric part counts
        for (int i = 0; i < j; ++i) {
            const double v = 0.5 * (Sm[(size_t)j * m + i] + Sm[(size_t)i * m + j]);
            Sm[(size_t)j * m + i] = Sm[(size_t)i * m + j] = v;
        }
    h->hS = Sm;
    h->useS = (Rm[0] != 0.0 && Sm[0] != 0.0) ? 1 : 0;  // the reference drops the S term together with R (src/sub/design_mpc.jl:423-466)
    h->rho = rho; h->sigma = sigma;

    // ---- state rows (state box, terminal equality): row tables and the shared constraint-space matrix
    { const int rc_ = setup_state_rows(h, xmin, xmax, false); if (rc_ != ALMPC_OK) return rc_; }
    std::vector<int> rowsel;   // rows of Gamma the state rows are, in row order
    for (int k = 0; k < N; ++k)
        for (int i = 0; i < n; ++i)
            if (h->has_box || (h->terminal_eq && k == N - 1)) rowsel.push_back(k * n + i);
    h->s0_basis_ok = false;
    if (!rowsel.empty() && !h->dVsPlain) HIP_TRY(h, dalloc(&h->dVsPlain, (size_t)n * h->nzs));
    if (h->nzs <= 64 && !h->dPlain) HIP_TRY(h, dalloc(&h->dPlain, 2 * (size_t)h->nz * h->nzs + 2 * (size_t)n * h->nzs));
    int rc = design_shared_device(h->stream, n, m, N, h->nzs, h->nrb, h->ks, h->ksf, Am, Bm, Qm, Rm, Sm, Pm, rho, sigma,
                                  h->dMinvFrag, h->dVFrag, h->dHFrag, h->dFFrag, h->dG, h->dD, h->H, h->F, h->d, h->err,
                                  rowsel, h->Rs, h->dGhat, h->dGnorm, h->rho_mode, h->dRho, rowsel.empty() ? nullptr : h->dVsPlain,
                                  h->nzs <= 64 ? h->dPlain : nullptr);
    if (rc != ALMPC_OK) return rc;
    if (h->terminal_eq && h->mc >= n && !getenv("ALMPC_NO_EQ_PROJECTION")) {
        // the n terminal-equality rows (the last n state rows) are in every working set: eliminate them here, once
        const int ne = n, eq0 = h->R - n, Rs = h->Rs;
        std::vector<double> GE((size_t)ne * Rs), GEE((size_t)ne * ne), Y;
        HIP_TRY(h, hipMemcpy(GE.data(), h->dGhat + (size_t)eq0 * Rs, GE.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int a = 0; a < ne; ++a)
            for (int b = 0; b < ne; ++b) GEE[(size_t)b * ne + a] = GE[(size_t)a * Rs + eq0 + b];
        hm::mat Winv = hm::eye(ne);
        bool ok = hm::lu_solve(GEE, Winv, ne, ne);
        for (double v : Winv) ok = ok && std::isfinite(v);
        if (ok) {   // (a singular Ghat_EE -- terminal state not reachable in N steps -- keeps the row-by-row path, which reports it)
            Y.assign((size_t)ne * Rs, 0.0);   // Y = Winv GhatE, row-major [ne][Rs]
            for (int a = 0; a < ne; ++a)
                for (int e = 0; e < ne; ++e) {
                    const double wv = Winv[(size_t)e * ne + a];   // Winv(a, e)
                    for (int b = 0; b < Rs; ++b) Y[(size_t)a * Rs + b] += wv * GE[(size_t)e * Rs + b];
                }
            std::vector<double> Wrow((size_t)ne * ne);
            for (int a = 0; a < ne; ++a)
                for (int e = 0; e < ne; ++e) Wrow[(size_t)a * ne + e] = Winv[(size_t)e * ne + a];
            double* dY = nullptr;
            HIP_TRY(h, dalloc(&h->dGhatE, GE.size())); HIP_TRY(h, dalloc(&h->dWinvE, Wrow.size())); HIP_TRY(h, dalloc(&dY, Y.size()));
            HIP_TRY(h, hipMemcpy(h->dGhatE, GE.data(), GE.size() * sizeof(double), hipMemcpyHostToDevice));
            HIP_TRY(h, hipMemcpy(h->dWinvE, Wrow.data(), Wrow.size() * sizeof(double), hipMemcpyHostToDevice));
            HIP_TRY(h, hipMemcpy(dY, Y.data(), Y.size() * sizeof(double), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_ghat_project, dim3(512), dim3(256), 0, h->stream, h->R, Rs, ne, eq0, h->dGhatE, dY, h->dGhat, h->dGnorm);
            HIP_TRY(h, hipGetLastError());
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            (void)hipFree(dY);
            h->eq_proj = 1;
        }
    }
    HIP_TRY(h, hipMemcpy(h->dUmin, umin, m * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->dUmax, umax, m * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->dA, A, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->dB, B, (size_t)n * m * sizeof(double), hipMemcpyHostToDevice));
    {
        // blocked rollout: s stages per block with one lane per (stage, state row); lane (j, i) holds row i of
        // [A^j B, ..., A B, B (stages t = 0..j), 0 ... | A^(j+1)]  (design-time host math, n x n products)
        int sblk = std::min(N, std::min(64 / n, ROLL_SMX / m));
        h->roll_s = 0;
        if (sblk >= 1 && n <= ROLL_NX && !getenv("ALMPC_ROLLOUT_STAGEWISE")) {
            std::vector<double> M((size_t)(ROLL_SMX + ROLL_NX) * 64, 0.0);
            std::vector<hm::mat> Apow(sblk + 1), ApB(sblk);  // A^j, A^j B
            Apow[0] = hm::eye(n);
            for (int j = 1; j <= sblk; ++j) Apow[j] = hm::mul(Am, Apow[j - 1], n, n, n);
            for (int j = 0; j < sblk; ++j) ApB[j] = hm::mul(Apow[j], Bm, n, n, m);
            for (int j = 0; j < sblk; ++j)
                for (int i = 0; i < n; ++i) {
                    const int lane = j * n + i;
                    for (int t = 0; t <= j; ++t)
                        for (int a = 0; a < m; ++a) M[(size_t)(t * m + a) * 64 + lane] = ApB[j - t][(size_t)a * n + i];
                    for (int c = 0; c < n; ++c) M[(size_t)(ROLL_SMX + c) * 64 + lane] = Apow[j + 1][(size_t)c * n + i];
                }
            if (h->dRollM) { (void)hipFree(h->dRollM); h->dRollM = nullptr; }
            HIP_TRY(h, dalloc(&h->dRollM, M.size()));
            HIP_TRY(h, hipMemcpy(h->dRollM, M.data(), M.size() * sizeof(double), hipMemcpyHostToDevice));
            h->roll_s = sblk; h->roll_nb = (N + sblk - 1) / sblk;
        }
    }
    // redo of the instances a step leaves without a certificate (default on): the stage-wise dual active set (k_sdual: also state rows
    // and S), and behind it -- input box only, S = 0 -- the primal Riccati active set
    h->sd.ready = false;
    if (h->fallback) {
        const bool useS = h->useS != 0;
        if (sdual_shape_ok(n, m, N, useS)) {
            hm::mat Rb = Rm;
            if (Rm[0] == 0.0) std::fill(Rb.begin(), Rb.end(), 0.0);
            const int rc_ = sdual_setup_shared(h, Am, Bm, Qm, Rb, useS ? &Sm : nullptr, Pm, xmin, xmax, h->terminal_eq != 0);
            if (rc_ != ALMPC_OK && h->fallback == 1) return rc_;
        } else if (h->fallback == 1 && (h->mc > 0 || useS))
            return fail(h, ALMPC_ERR_UNSUPPORTED, "structured fallback: state rows / input-rate weight need n + m <= 48 and (N + 1)(n + m) <= 4096");
        if (riccati_shape_ok(h) && h->mc == 0 && !useS) {
            const int rc_ = riccati_weights(h, Qm, Rm, Pm.data());
            if (rc_ != ALMPC_OK) return rc_;
            h->r_batched_P = false; h->rP_stride = 0;
        }
    }
    h->designed = true;
    h->batched = false;
    h->ltv = false;
    // default references: zeros, shared
    std::vector<double> xr((size_t)n * (N + 1), 0.0), ur((size_t)nz, 0.0);
    const int rc_ref = almpc_set_reference(h, xr.data(), ur.data(), 0);
    if (rc_ref != ALMPC_OK) h->designed = false;
    return rc_ref;
}

// Per-instance models: every instance gets its own condensed QP from (A_i, B_i).  The design kernels of
// almpc_design.hip.h run with blockIdx.y = instance; the per-step path is k_admm_inst + k_polish<false> with strides.
}  // extern "C"

namespace {
// Time-varying inputs of almpc_design_ltv (host pointers); nullptr for the time-invariant almpc_design_batched.
struct LtvInputs {
    const double* A_all;   // [batch][N][n*n]
    const double* B_all;   // [batch][N][n*m]
    const double* c_all;   // [batch][N][n] or null
    const double* ebar;    // [batch][N][n]  xbar_{k+1} - x_ref_{k+1}
    const double* qadd;    // [batch][nz]    input part of the gradient
    const double* ubar;    // [batch][nz]    linearisation inputs (become the per-instance input reference)
    const double* xbar;    // [batch][(N+1)*n]
};

// Persistent per-instance operands of the batched / LTV / SQP designs (allocated once per handle).
int ensure_batched_alloc(almpc_handle* h) {
    const int n = h->n, m = h->m, N = h->N, nz = h->nz, nzs = h->nzs;
    const size_t b = (size_t)h->batch;
    const int njf = (n + 15) / 16, ps = 16 * njf, gs = nzs;
    const int kr = ((n * N + HESS_KC - 1) / HESS_KC) * HESS_KC;
    if (!h->batched_alloc) {
        HIP_TRY(h, dalloc(&h->bA, b * n * n)); HIP_TRY(h, dalloc(&h->bB, b * n * m));
        HIP_TRY(h, dalloc(&h->bMinv, b * nz * nzs)); HIP_TRY(h, dalloc(&h->bG, b * nz * nzs)); HIP_TRY(h, dalloc(&h->bHs, b * nz * nzs));
        HIP_TRY(h, dalloc(&h->bFs, b * n * nzs)); HIP_TRY(h, dalloc(&h->bVs, b * n * nzs));
        HIP_TRY(h, dalloc(&h->bD, b * nzs)); HIP_TRY(h, dalloc(&h->bRho, b * nzs));
        HIP_TRY(h, dalloc(&h->bH, b * nz * nz)); HIP_TRY(h, dalloc(&h->bF, b * nz * n));
        if (design_instance_lds_doubles(n, m, N) * sizeof(double) > 160 * 1024) {  // dense route only: Gamma panels in HBM
            HIP_TRY(h, dalloc(&h->bPhi, b * N * n * n)); HIP_TRY(h, dalloc(&h->bGk, b * N * n * m));
            HIP_TRY(h, dalloc(&h->bGam, b * kr * gs)); HIP_TRY(h, dalloc(&h->bW, b * kr * gs)); HIP_TRY(h, dalloc(&h->bWP, b * kr * ps));
            // padding rows / columns of the row-major panels stay zero for the lifetime of the handle
            HIP_TRY(h, hipMemset(h->bGam, 0, b * kr * gs * sizeof(double)));
            HIP_TRY(h, hipMemset(h->bW, 0, b * kr * gs * sizeof(double)));
            HIP_TRY(h, hipMemset(h->bWP, 0, b * kr * ps * sizeof(double)));
        }
        HIP_TRY(h, dalloc(&h->bP, b * n * n));
        HIP_TRY(h, dalloc(&h->bFlag, b));
        HIP_TRY(h, hipMemset(h->bHs, 0, b * nz * nzs * sizeof(double)));
        HIP_TRY(h, hipMemset(h->bFs, 0, b * n * nzs * sizeof(double)));
        HIP_TRY(h, hipMemset(h->bVs, 0, b * n * nzs * sizeof(double)));
        HIP_TRY(h, hipMemset(h->bMinv, 0, b * nz * nzs * sizeof(double)));
        HIP_TRY(h, hipMemset(h->bG, 0, b * nz * nzs * sizeof(double)));
        h->batched_alloc = true;
    }
    return ALMPC_OK;
}

// H_i (column-major, in bH) -> Jacobi scaling d_i, H'_i, F'_i, G_i = H'_i^-1, rho_i, Minv_i = (H'_i + sigma I + diag(rho_i))^-1
// (scaled: the producer of H_i has done the scaling as its own tail)
// (with_v: also V_i = -G_i F'_i -- inside the first inverse's launch where that kernel can, else by launch_neg_gm_batched)
// (v1M / v1Out: instead of V_i, ONE column per instance -- the SQP iteration's v0S_i = -G_i fS_i, [batch][nz] each)
void launch_batched_factor(almpc_handle* h, const DesignStrides& ds, double rho, double sigma, hipStream_t st, bool scaled = false, bool with_v = false,
                           const double* v1M = nullptr, double* v1Out = nullptr) {
    const int n = h->n, nz = h->nz, nzs = h->nzs;
    const unsigned gb = (unsigned)h->batch;
    if (!scaled) hipLaunchKernelGGL(k_design_scale, dim3(1, gb), dim3(256), 0, st, nz, nzs, n, h->bH, h->bF, h->bD, h->bHs, h->bFs, h->bFlag, ds);
    const size_t inv_lds = 520 * sizeof(double);
    if (!v1M && !h->skip_admm && h->rho_mode == 0 && design_inverse_makes_rho(nz, nzs) && nz <= 64 && design_inverse_makes_v(nz) && !getenv("ALMPC_DBG_SPLIT_INVERSES")) {
        // scalar rho: the ADMM's KKT inverse does not need G_i -- both inverses, the penalty profile and V_i in ONE launch
        h->minv_packed = false;
        launch_design_inverse(dim3(1, gb), inv_lds, st, nz, nzs, h->bHs, 0.0, (const double*)nullptr, h->bG, h->bFlag, ds.Hs, ds.rho, ds.G, 1L,
                              (const double*)nullptr, 0L, 0, rho, h->bRho, with_v ? h->bFs : nullptr, h->bVs, ds.Fs, n, h->bMinv, ds.Minv, sigma);
        return;
    }
    if (v1M && design_inverse_makes_v(nz))   // one column per instance from the rows / columns of G_i the inverse still holds
        launch_design_inverse(dim3(1, gb), inv_lds, st, nz, nzs, h->bHs, 0.0, (const double*)nullptr, h->bG, h->bFlag, ds.Hs, 0L, ds.G, 1L,
                              (const double*)nullptr, 0L, 0, 0.0, (double*)nullptr, v1M, v1Out, (long)nz, 1, (double*)nullptr, 0L, 0.0, nz);
    else if (with_v && nz <= 64 && design_inverse_makes_v(nz))   // V_i = -G_i F'_i from the rows of G_i the inverse's wave still holds (the
                                                                  // column-split kernel: measured slower than k_neg_gm_cols for n columns)
        launch_design_inverse(dim3(1, gb), inv_lds, st, nz, nzs, h->bHs, 0.0, (const double*)nullptr, h->bG, h->bFlag, ds.Hs, 0L, ds.G, 1L,
                              (const double*)nullptr, 0L, 0, 0.0, (double*)nullptr, h->bFs, h->bVs, ds.Fs, n);
    else {
        launch_design_inverse(dim3(1, gb), inv_lds, st, nz, nzs, h->bHs, 0.0, (const double*)nullptr, h->bG, h->bFlag, ds.Hs, 0L, ds.G, 1L);
        if (with_v) launch_neg_gm_batched(st, gb, nz, nzs, n, h->bG, h->bFs, h->bVs, ds.G, ds.Fs);
    }
    if (h->skip_admm) return;  // no ADMM phase in this solve: its KKT inverse is not needed
    h->minv_packed = design_inverse_can_pack(nz);   // the ADMM's KKT inverse as its packed triangle: half the stream of k_admm_inst
    const long sMinv = h->minv_packed ? packed_tri_doubles(nz) : ds.Minv;
    if (design_inverse_makes_rho(nz, nzs))   // the penalty profile is made inside the inverse's own launch
        launch_design_inverse(dim3(1, gb), inv_lds, st, nz, nzs, h->bHs, sigma, (const double*)nullptr, h->bMinv, h->bFlag, ds.Hs, ds.rho, ds.Minv, 1L,
                              h->bG, ds.G, h->rho_mode, rho, h->bRho);
    else {
        hipLaunchKernelGGL(k_design_rho, dim3(1, gb), dim3(256), 0, st, nz, nzs, h->rho_mode, rho, h->bG, h->bRho, ds.G, ds.rho);
        launch_design_inverse(dim3(1, gb), inv_lds, st, nz, nzs, h->bHs, sigma, (const double*)h->bRho, h->bMinv, h->bFlag, ds.Hs, ds.rho, sMinv, 1L,
                              (const double*)nullptr, 0L, 0, 0.0, (double*)nullptr, (const double*)nullptr, (double*)nullptr, 0L, 0, (double*)nullptr, 0L, 0.0, 0,
                              h->minv_packed ? 1 : 0);
    }
}

DesignStrides batched_strides(const almpc_handle* h, bool p_inst) {
    const int n = h->n, m = h->m, N = h->N, nz = h->nz, nzs = h->nzs;
    const int njf = (n + 15) / 16, ps = 16 * njf, gs = nzs;
    const int kr = ((n * N + HESS_KC - 1) / HESS_KC) * HESS_KC;
    DesignStrides ds;
    ds.A = (long)n * n; ds.B = (long)n * m; ds.P = p_inst ? (long)n * n : 0; ds.Phi = (long)N * n * n; ds.Gk = (long)N * n * m;
    ds.Gam = (long)kr * gs; ds.WP = (long)kr * ps; ds.H = (long)nz * nz; ds.F = (long)nz * n; ds.d = nzs; ds.Hs = (long)nz * nzs;
    ds.Fs = (long)n * nzs; ds.G = (long)nz * nzs; ds.Minv = (long)nz * nzs; ds.rho = nzs; ds.flag = 1;
    return ds;
}

// Jacobians of an Fnn at p.batch points: wave-per-point build when weights + 4 waves' buffers fit 64 KB of LDS, else one workgroup
// per point.
hipError_t launch_fnn_jacobian(const FnnParams& p, int num_cus, hipStream_t st) {
    const size_t nin = (size_t)p.n + p.m;
    // small networks (H (n + m) <= 192 entries of the Jacobian being propagated): two points per wave, one per half-wave
    // (four points per wave, a quarter-wave each: measured no better, 0.305 against 0.303 ms per SQP iteration)
    const int ppw = ((size_t)p.H * nin <= 192 && !getenv("ALMPC_FNN_ONE_POINT_PER_WAVE")) ? 2 : 1;
    const size_t lw = fnn_w_lds_doubles(p.n, p.m, p.H, p.L, ppw) * sizeof(double);
    if (lw <= 64 * 1024 && !getenv("ALMPC_FNN_WG")) {
        int wgs = (p.batch + FNN_W_WAVES * ppw - 1) / (FNN_W_WAVES * ppw);
        const int cap = num_cus * 8;  // 32 waves per CU: a point is a latency chain on one wave (18 us); 16 waves per CU took 55 us for the
                                      // 12800 points of an SQP iteration, 32 take 48 (52 workgroups per CU: 49.5)
        if (wgs > cap) wgs = cap;
        if (ppw == 2) hipLaunchKernelGGL((k_fnn_jacobian_w<32>), dim3(wgs), dim3(64 * FNN_W_WAVES), lw, st, p);
        else hipLaunchKernelGGL((k_fnn_jacobian_w<64>), dim3(wgs), dim3(64 * FNN_W_WAVES), lw, st, p);
        return hipGetLastError();
    }
    const size_t lds = (2 * (size_t)p.H + 2 * (size_t)p.H * nin + nin) * sizeof(double);
    if (lds > 64 * 1024) {
        const hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(k_fnn_jacobian), (size_t)(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_fnn_jacobian, dim3(p.batch), dim3(256), lds, st, p);
    return hipGetLastError();
}

// k_design_ltv: accumulators in registers when the shape allows (nz <= 128, one thread per element of (A, B)), else in LDS
bool ltv_reg_path(const almpc_handle* h) {
    const int n = h->n, m = h->m;
    return h->nz <= LTV_REG_NZ && n * n + n * m <= 1024 - n && design_ltv_reg_lds_doubles(n, m, h->N) * sizeof(double) <= 160 * 1024 &&
           !getenv("ALMPC_LTV_LDS");
}

bool ltv_supported(const almpc_handle* h) {
    return ltv_reg_path(h) || design_ltv_lds_doubles(h->n, h->m, h->N) * sizeof(double) <= 160 * 1024;
}

template <int NC>
hipError_t launch_design_ltv_reg(almpc_handle* h, const DesignLtvParams& lp, size_t lds, hipStream_t st) {
    // 8 x 8 register tiles on 256 threads (ALMPC_LTV_T8=1): measured SLOWER (SQP iteration 0.392 against 0.348 ms): the stage is bound
    // by the latency of its dependent LDS steps, which 16 waves hide better than four, not by LDS bytes
    const bool t8 = NC > 0 && h->n * h->n + h->n * h->m <= 256 - h->n && getenv("ALMPC_LTV_T8");
    if (t8) {
        if (lds > 64 * 1024) {
            const hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(k_design_ltv_reg<NC, 8>), (size_t)(lds));
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((k_design_ltv_reg<NC, 8>), dim3((unsigned)h->batch), dim3(256), lds, st, lp);
        return hipGetLastError();
    }
    if (lds > 64 * 1024) {
        const hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(k_design_ltv_reg<NC, 4>), (size_t)(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_design_ltv_reg<NC, 4>), dim3((unsigned)h->batch), dim3(1024), lds, st, lp);
    return hipGetLastError();
}

hipError_t launch_design_ltv(almpc_handle* h, const DesignLtvParams& lp, hipStream_t st) {
    const int n = h->n, m = h->m;
    if (ltv_reg_path(h)) {
        const size_t lds = design_ltv_reg_lds_doubles(n, m, h->N) * sizeof(double);
        switch (n) {
            case 2: return launch_design_ltv_reg<2>(h, lp, lds, st);
            case 4: return launch_design_ltv_reg<4>(h, lp, lds, st);
            case 6: return launch_design_ltv_reg<6>(h, lp, lds, st);
            case 12: return launch_design_ltv_reg<12>(h, lp, lds, st);
            default: return launch_design_ltv_reg<0>(h, lp, lds, st);
        }
    }
    const size_t lds = design_ltv_lds_doubles(n, m, h->N) * sizeof(double);
    if (lds > 64 * 1024) {
        const hipError_t e = ensure_dyn_lds(reinterpret_cast<const void*>(k_design_ltv), (size_t)(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_design_ltv, dim3((unsigned)h->batch), dim3(256), lds, st, lp);
    return hipGetLastError();
}

// whether k_design_instance_t can linearise the network itself (LDS route with room for the weights and one wave's scratch)
bool design_fuses_fnn(const almpc_handle* h, int H, int L) {
    const size_t lds = (design_instance_lds_doubles(h->n, h->m, h->N) + fnn_weights_doubles(h->n, h->m, H, L) + fnn_wave_scratch_doubles(h->n, h->m, H)) * sizeof(double);
    return lds <= 160 * 1024 && !getenv("ALMPC_DBG_SPLIT_JACOBIAN");
}

// The per-instance design from DEVICE-resident operands (bA, bB, bP; weights dQ, dR, dS): prediction matrices, H_i and F_i,
// scaling, both inverses, V_i.  Launches only (handle's stream); the caller checks bFlag.
// fuse_fnn: the models are linearisations of this network at fuse_fnn->x / u (re-linearisation pipeline): done by the design kernel's
// own workgroups when *fused comes back true -- else the caller launches the Jacobians first
hipError_t launch_batched_design(almpc_handle* h, const DesignStrides& ds, int useR, int useS, const double* dQ, const double* dR,
                                 const double* dS, double rho, double sigma, const FnnParams* fuse_fnn = nullptr) {
    const int n = h->n, m = h->m, N = h->N, nz = h->nz, nzs = h->nzs, nrb = h->nrb;
    const int njf = (n + 15) / 16, ps = 16 * njf, gs = nzs;
    const int kr = ((n * N + HESS_KC - 1) / HESS_KC) * HESS_KC;
    const unsigned gb = (unsigned)h->batch;
    hipStream_t st = h->stream;
    size_t inst_lds = design_instance_lds_doubles(n, m, N) * sizeof(double);
    hipError_t e = hipSuccess;
    size_t fnn_off = 0;
    if (fuse_fnn) {   // (the caller has checked design_fuses_fnn)
        fnn_off = design_instance_lds_doubles(n, m, N);
        inst_lds += (fnn_weights_doubles(n, m, fuse_fnn->H, fuse_fnn->L) + fnn_wave_scratch_doubles(n, m, fuse_fnn->H)) * sizeof(double);
    }
    if (inst_lds > 160 * 1024) {   // (the LDS route below clears the flags itself)
        e = hipMemsetAsync(h->bFlag, 0, (size_t)h->batch * sizeof(int), st);
        if (e != hipSuccess) return e;
    }
    bool scaled = false;
    if (inst_lds <= 160 * 1024) {  // structured route: H_i, F_i from the Toeplitz blocks in LDS, no Gamma panels in HBM
        DesignInstParams dp;
        dp.flag = h->bFlag; dp.sFlag = 1;
        dp.n = n; dp.m = m; dp.N = N; dp.nz = nz; dp.useR = useR; dp.useS = useS;
        dp.A = h->bA; dp.B = h->bB; dp.P = h->bP; dp.sA = ds.A; dp.sB = ds.B; dp.sP = ds.P;
        dp.Q = dQ; dp.R = dR; dp.S = dS; dp.H = h->bH; dp.F = h->bF; dp.sH = ds.H; dp.sF = ds.F;
        if (nz <= 128 && design_instance_lds_doubles(n, m, N) >= 128 && !getenv("ALMPC_DBG_SPLIT_SCALE")) {   // the scaling rides along (k_design_scale's body, one launch less)
            dp.d = h->bD; dp.Hs = h->bHs; dp.Fs = h->bFs; dp.sd = ds.d; dp.sHs = ds.Hs; dp.sFs = ds.Fs; dp.nzs = nzs;
        }
        scaled = dp.Hs != nullptr;
        if (fuse_fnn) { dp.fnn_on = 1; dp.fnn_off = fnn_off; dp.fnn = *fuse_fnn; }
#define DESIGN_INST(NC_, MC_)                                                                                \
    do {                                                                                                     \
        e = ensure_dyn_lds(reinterpret_cast<const void*>(k_design_instance_t<NC_, MC_>), inst_lds);          \
        if (e != hipSuccess) return e;                                                                       \
        hipLaunchKernelGGL((k_design_instance_t<NC_, MC_>), dim3(gb), dim3(256), inst_lds, st, dp);          \
    } while (0)
        if (n == 12 && m == 4) DESIGN_INST(12, 4);
        else if (n == 4 && m == 2) DESIGN_INST(4, 2);
        else if (n == 2 && m == 1) DESIGN_INST(2, 1);
        else DESIGN_INST(0, 0);
#undef DESIGN_INST
    } else {
        hipLaunchKernelGGL(k_design_blocks, dim3(1, gb), dim3(256), (size_t)(3 * n * n + n * m) * sizeof(double), st, n, m, N,
                           h->bA, h->bB, h->bPhi, h->bGk, ds);
        hipLaunchKernelGGL(k_design_gamma, dim3(N, gb), dim3(256), 0, st, n, m, N, dQ, h->bP, h->bPhi, h->bGk, h->bGam, h->bW, h->bWP, gs, ps, ds);
        HessParams hp;
        hp.n = n; hp.m = m; hp.N = N; hp.nz = nz; hp.nrb = nrb; hp.njf = njf; hp.kr = kr;
        hp.Gam = h->bGam; hp.W = h->bW; hp.WP = h->bWP; hp.gs = gs; hp.ps = ps; hp.R = dR; hp.S = dS; hp.useR = useR; hp.useS = useS;
        hp.H = h->bH; hp.F = h->bF; hp.st = ds;
        const size_t hess_lds = (size_t)HESS_KC * (16 + (gs + 16) + (ps + 16)) * sizeof(double);
        hipLaunchKernelGGL(k_design_hessian, dim3(nrb, gb), dim3(64 * (nrb + njf)), hess_lds, st, hp);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    DesignStrides ds2 = ds;
    ds2.h_symmetric = (inst_lds <= 160 * 1024) ? 1 : 0;   // (the LDS route writes both halves of H_i from one value)
    launch_batched_factor(h, ds2, rho, sigma, st, scaled, true);
    return hipGetLastError();
}

int design_batched_common(almpc_handle* h, const double* A_batch, const double* B_batch, const double* Q, const double* R,
                          const double* S, const double* P, int P_per_instance, const double* umin, const double* umax,
                          double rho, double sigma, const LtvInputs* ltv) {
    if (!h) return ALMPC_ERR_INVALID;
    drop_lazy_redo(h);
    if (h->structured) {   // one model per instance, structured solve: models and terminal weights on the device, nothing condensed
        if (ltv) return fail(h, ALMPC_ERR_UNSUPPORTED, "structured solve: time-varying designs go through almpc_sqp_fnn_* (stage models on the device)");
        if (!A_batch || !B_batch || !Q || !R || !umin || !umax) return fail(h, ALMPC_ERR_INVALID, "design_batched: null matrix pointer");
        const int n = h->n, m = h->m, N = h->N;
        const size_t b = (size_t)h->batch;
        const bool useS = R[0] != 0.0 && S && S[0] != 0.0;
        h->designed = false;
        HIP_TRY(h, hipSetDevice(h->device));
        hm::mat Qm(Q, Q + (size_t)n * n), Rm(R, R + (size_t)m * m), Sm;
        if (useS) {
            Sm.assign(S, S + (size_t)m * m);
            for (int j = 0; j < m; ++j)
                for (int i = 0; i < j; ++i) { const double v = 0.5 * (Sm[(size_t)j * m + i] + Sm[(size_t)i * m + j]); Sm[(size_t)j * m + i] = Sm[(size_t)i * m + j] = v; }
        }
        const bool p_inst = P ? (P_per_instance != 0) : true;
        hm::mat Pall;
        if (P) Pall.assign(P, P + (p_inst ? b : 1) * (size_t)n * n);
        else {
            Pall.resize(b * (size_t)n * n);
            for (size_t i = 0; i < b; ++i) {
                hm::mat Am(A_batch + i * n * n, A_batch + (i + 1) * n * n), Bm(B_batch + i * n * m, B_batch + (i + 1) * n * m), Pm;
                if (!hm::dare(Am, Bm, Qm, Rm, n, m, Pm)) return fail(h, ALMPC_ERR_NUMERIC, "design_batched: DARE did not converge for instance " + std::to_string(i));
                std::copy(Pm.begin(), Pm.end(), Pall.begin() + i * n * n);
            }
        }
        if (!h->bA) HIP_TRY(h, dalloc(&h->bA, b * n * n));
        if (!h->bB) HIP_TRY(h, dalloc(&h->bB, b * n * m));
        if (!h->bP) HIP_TRY(h, dalloc(&h->bP, b * n * n));
        HIP_TRY(h, hipMemcpy(h->bA, A_batch, b * n * n * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->bB, B_batch, b * n * m * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->bP, Pall.data(), Pall.size() * sizeof(double), hipMemcpyHostToDevice));
        h->batched = true; h->ltv = false; h->r_batched_P = true; h->rP_stride = p_inst ? (long)n * n : 0;
        const double* bxmin = h->boxmin.empty() ? nullptr : h->boxmin.data();
        const double* bxmax = h->boxmax.empty() ? nullptr : h->boxmax.data();
        h->sd.ready = false;
        if (sdual_shape_ok(n, m, N, useS) && !getenv("ALMPC_STRUCTURED_PRIMAL")) {
            const int rc_ = sdual_setup_batched(h, Qm, Rm, useS ? &Sm : nullptr, P == nullptr && !useS, bxmin, bxmax, h->terminal_eq != 0);
            if (rc_ != ALMPC_OK) return rc_;
            HIP_TRY(h, launch_sgains(h, 0));
            std::vector<int> bad(b, 0);
            HIP_TRY(h, hipMemcpyAsync(bad.data(), h->sd.bad, b * sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            for (size_t i = 0; i < b; ++i)
                if (bad[i]) return fail(h, ALMPC_ERR_NUMERIC, "design_batched: R + B'PB is not positive definite for instance " + std::to_string(i));
        } else if (bxmin || h->terminal_eq || useS)
            return fail(h, ALMPC_ERR_UNSUPPORTED, "structured solve: state rows / input-rate weight need the stage-wise dual solve (n + m <= 48, (N + 1)(n + m) <= 4096)");
        if (riccati_shape_ok(h)) { const int rc_ = riccati_weights(h, Qm, Rm, nullptr); if (rc_ != ALMPC_OK) return rc_; }
        else if (!h->sd.ready) return fail(h, ALMPC_ERR_UNSUPPORTED, "structured solve: shape outside both stage-wise solvers");
        HIP_TRY(h, hipMemcpy(h->dUmin, umin, m * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->dUmax, umax, m * sizeof(double), hipMemcpyHostToDevice));
        h->P.assign(Pall.begin(), Pall.begin() + (size_t)n * n); h->H.clear(); h->F.clear(); h->d.clear();
        h->hS.assign((size_t)m * m, 0.0); h->useS = useS ? 1 : 0;
        if (useS) h->hS = Sm;
        h->has_box = bxmin ? 1 : 0;
        h->designed = true;
        std::vector<double> xr((size_t)n * (N + 1), 0.0), ur((size_t)h->nz, 0.0);
        const int rc_ref = almpc_set_reference(h, xr.data(), ur.data(), 0);
        if (rc_ref != ALMPC_OK) h->designed = false;
        return rc_ref;
    }
    if (!A_batch || !B_batch || !Q || !R || !umin || !umax) return fail(h, ALMPC_ERR_INVALID, "design_batched: null matrix pointer");
    if (!(rho > 0.0) || !(sigma >= 0.0)) return fail(h, ALMPC_ERR_INVALID, "design_batched: rho must be > 0 and sigma >= 0");
    h->designed = false;  // the previous design is overwritten below; set again by the last statement on success
    h->sqp.ready = h->sqp.started = false;
    h->relin.ready = false;
    // ALMPC_DESIGN_TRACE=1: host-side time between the marks below on stderr (where a design call spends its wall clock)
    const bool trace_ = getenv("ALMPC_DESIGN_TRACE") != nullptr;
    auto t_mark_ = std::chrono::steady_clock::now();
    auto tr = [&](const char* what) {
        if (!trace_) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[almpc design] %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(now - t_mark_).count());
        t_mark_ = now;
    };
    HIP_TRY(h, hipSetDevice(h->device));
    const int n = h->n, m = h->m, N = h->N, nz = h->nz, nzs = h->nzs;
    const size_t b = (size_t)h->batch;
    for (int i = 0; i < m; ++i)
        if (!(umin[i] <= umax[i])) return fail(h, ALMPC_ERR_INVALID, "design_batched: umin > umax");
    hm::mat Qm(Q, Q + (size_t)n * n), Rm(R, R + (size_t)m * m);
    hm::mat Sm = S ? hm::mat(S, S + (size_t)m * m) : hm::mat((size_t)m * m, 0.0);
    auto symmetrise = [](hm::mat& M, int k) {
        for (int j = 0; j < k; ++j)
            for (int i = 0; i < j; ++i) {
                const double v = 0.5 * (M[(size_t)j * k + i] + M[(size_t)i * k + j]);
                M[(size_t)j * k + i] = M[(size_t)i * k + j] = v;
            }
    };
    symmetrise(Qm, n); symmetrise(Rm, m); symmetrise(Sm, m);
    // terminal weight: given (shared or per instance), else DARE of every instance's model (src/sub/design_mpc.jl:327)
    const bool p_inst = P ? (P_per_instance != 0) : true;
    hm::mat Pall;
    if (P) Pall.assign(P, P + (p_inst ? b : 1) * (size_t)n * n);
    else {
        Pall.resize(b * (size_t)n * n);
        for (size_t i = 0; i < b; ++i) {
            hm::mat Am(A_batch + i * n * n, A_batch + (i + 1) * n * n), Bm(B_batch + i * n * m, B_batch + (i + 1) * n * m), Pm;
            if (!hm::dare(Am, Bm, Qm, Rm, n, m, Pm)) return fail(h, ALMPC_ERR_NUMERIC, "design_batched: DARE did not converge for instance " + std::to_string(i));
            std::copy(Pm.begin(), Pm.end(), Pall.begin() + i * n * n);
        }
    }
    for (size_t i = 0; i < (p_inst ? b : 1); ++i) {
        hm::mat Pm(Pall.begin() + i * n * n, Pall.begin() + (i + 1) * n * n);
        symmetrise(Pm, n);
        std::copy(Pm.begin(), Pm.end(), Pall.begin() + i * n * n);
    }
    h->P.assign(Pall.begin(), Pall.begin() + (size_t)n * n);
    h->hS = Sm;
    h->rho = rho; h->sigma = sigma;
    const int useR = Rm[0] != 0.0, useS = useR && Sm[0] != 0.0;
    h->useS = useS;
    tr("host preparation");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    tr("stream idle");
    // state rows (almpc_set_state_box, almpc_set_terminal_equality): one constraint-space matrix per instance
    { const int rc_ = setup_state_rows(h, h->boxmin.empty() ? nullptr : h->boxmin.data(), h->boxmax.empty() ? nullptr : h->boxmax.data(), true);
      if (rc_ != ALMPC_OK) return rc_; }
    { const int rc_ = ensure_batched_alloc(h); if (rc_ != ALMPC_OK) return rc_; }
    hipStream_t st = h->stream;
    tr("state rows + allocation");
    if (!h->wQ) HIP_TRY(h, dalloc(&h->wQ, (size_t)n * n));
    if (!h->wR) HIP_TRY(h, dalloc(&h->wR, (size_t)m * m));
    if (!h->wS) HIP_TRY(h, dalloc(&h->wS, (size_t)m * m));
    double *dQ = h->wQ, *dR = h->wR, *dS = h->wS;   // (kept with the handle: three hipMalloc + hipFree per design cost 0.15 ms)
    auto release = [&]() {};
#define BTRY(call)                                                                                                  \
    do {                                                                                                            \
        hipError_t e_ = (call);                                                                                     \
        if (e_ != hipSuccess) { release(); return fail(h, ALMPC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } \
    } while (0)
    BTRY(hipMemcpyAsync(h->bA, A_batch, b * n * n * sizeof(double), hipMemcpyHostToDevice, st));
    BTRY(hipMemcpyAsync(h->bB, B_batch, b * n * m * sizeof(double), hipMemcpyHostToDevice, st));
    BTRY(hipMemcpyAsync(h->bP, Pall.data(), Pall.size() * sizeof(double), hipMemcpyHostToDevice, st));
    BTRY(hipMemcpyAsync(dQ, Qm.data(), Qm.size() * sizeof(double), hipMemcpyHostToDevice, st));
    BTRY(hipMemcpyAsync(dR, Rm.data(), Rm.size() * sizeof(double), hipMemcpyHostToDevice, st));
    BTRY(hipMemcpyAsync(dS, Sm.data(), Sm.size() * sizeof(double), hipMemcpyHostToDevice, st));
    BTRY(hipMemsetAsync(h->bFlag, 0, b * sizeof(int), st));
    tr("weights alloc + copies queued");
    const DesignStrides ds = batched_strides(h, p_inst);
    const unsigned gb = (unsigned)b;
    double *dAll = nullptr, *dBll = nullptr, *dC = nullptr, *dE = nullptr, *dQa = nullptr;  // LTV staging (freed below)
    if (ltv) {
        auto upl = [&](double** d, const double* src, size_t cnt) -> hipError_t {
            hipError_t e = dalloc(d, cnt);
            if (e == hipSuccess) e = hipMemcpyAsync(*d, src, cnt * sizeof(double), hipMemcpyHostToDevice, st);
            return e;
        };
        hipError_t e = upl(&dAll, ltv->A_all, b * N * n * n);
        if (e == hipSuccess) e = upl(&dBll, ltv->B_all, b * N * n * m);
        if (e == hipSuccess && ltv->c_all) e = upl(&dC, ltv->c_all, b * N * n);
        if (e == hipSuccess) e = upl(&dE, ltv->ebar, b * N * n);
        if (e == hipSuccess) e = upl(&dQa, ltv->qadd, b * nz);
        if (e == hipSuccess && !h->bQ) e = dalloc(&h->bQ, b * nz);
        if (e == hipSuccess) e = hipMemsetAsync(h->bF, 0, b * nz * n * sizeof(double), st);
        if (e == hipSuccess) {
            DesignLtvParams lp;
            lp.n = n; lp.m = m; lp.N = N; lp.nz = nz; lp.useR = useR; lp.useS = useS;
            lp.A = dAll; lp.B = dBll; lp.c = dC; lp.ebar = dE; lp.P = h->bP; lp.sP = ds.P; lp.Q = dQ; lp.R = dR; lp.S = dS;
            lp.qadd = dQa; lp.H = h->bH; lp.q = h->bQ;
            e = launch_design_ltv(h, lp, st);
        }
        if (e == hipSuccess && h->mc > 0) {   // (needs G_i, d_i: the factor step comes first when there are state rows)
            launch_batched_factor(h, ds, rho, sigma, st);
            e = launch_ghat_inst(h, dAll, dBll);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);  // the staging buffers are released right away
        for (double** q : {&h->lA, &h->lB, &h->lC, &h->lE})
            if (*q) { (void)hipFree(*q); *q = nullptr; }
        if (e == hipSuccess && h->mc > 0) {   // ... except with state rows: the step rolls the stage models out
            h->lA = dAll; h->lB = dBll; h->lC = dC; h->lE = dE;
            dAll = dBll = dC = dE = nullptr;
        }
        (void)hipFree(dAll); (void)hipFree(dBll); (void)hipFree(dC); (void)hipFree(dE); (void)hipFree(dQa);
        if (e != hipSuccess) { release(); return fail(h, ALMPC_ERR_HIP, std::string("design_ltv: ") + hipGetErrorString(e)); }
        if (h->mc == 0) launch_batched_factor(h, ds, rho, sigma, st);
        launch_neg_gm_batched(st, gb, nz, nzs, n, h->bG, h->bFs, h->bVs, ds.G, ds.Fs);
        BTRY(hipGetLastError());
    } else {
        BTRY(launch_batched_design(h, ds, useR, useS, dQ, dR, dS, rho, sigma));
        if (h->mc > 0) BTRY(launch_ghat_inst(h, nullptr, nullptr));
    }
    tr("kernels queued");
    std::vector<int> flags(b, 0);
    BTRY(hipMemcpyAsync(flags.data(), h->bFlag, b * sizeof(int), hipMemcpyDeviceToHost, st));
    // host copies of instance 0 for almpc_get_design (every instance: almpc_get_design_instance)
    h->H.assign((size_t)nz * nz, 0.0); h->F.assign((size_t)nz * n, 0.0); h->d.assign((size_t)nzs, 0.0);
    BTRY(hipMemcpyAsync(h->H.data(), h->bH, h->H.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    BTRY(hipMemcpyAsync(h->F.data(), h->bF, h->F.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    BTRY(hipMemcpyAsync(h->d.data(), h->bD, h->d.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    BTRY(hipStreamSynchronize(st));
    tr("device done");
#undef BTRY
    release();
    tr("weights freed");
    for (size_t i = 0; i < b; ++i)
        if (flags[i] != 0)
            return fail(h, ALMPC_ERR_NUMERIC, "design_batched: instance " + std::to_string(i) +
                        (flags[i] == 1 ? ": condensed Hessian has a non-positive diagonal" : ": Cholesky pivot not positive"));
    HIP_TRY(h, hipMemcpy(h->dUmin, umin, m * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->dUmax, umax, m * sizeof(double), hipMemcpyHostToDevice));
    h->sd.ready = false;
    if (h->fallback && !ltv) {
        h->r_batched_P = true; h->rP_stride = p_inst ? (long)n * n : 0;
        if (sdual_shape_ok(n, m, N, useS != 0)) {
            // (one stage of records per instance when the terminal weight is the instance's own DARE solution)
            const int rc_ = sdual_setup_batched(h, Qm, Rm, useS ? &Sm : nullptr, P == nullptr && !useS,
                                                h->boxmin.empty() ? nullptr : h->boxmin.data(), h->boxmax.empty() ? nullptr : h->boxmax.data(), h->terminal_eq != 0);
            if (rc_ != ALMPC_OK && h->fallback == 1) return rc_;
            // (the stage records are computed when a step leaves instances to redo, for those instances only: nothing here)
        } else if (h->fallback == 1 && (h->mc > 0 || useS))
            return fail(h, ALMPC_ERR_UNSUPPORTED, "structured fallback: state rows / input-rate weight need n + m <= 48 and (N + 1)(n + m) <= 4096");
        if (riccati_shape_ok(h) && h->mc == 0 && !useS) {
            const int rc_ = riccati_weights(h, Qm, Rm, nullptr);
            if (rc_ != ALMPC_OK) return rc_;
        }
    }
    h->designed = true;
    h->batched = true;
    h->ltv = false;
    if (!ltv) {
        std::vector<double> xr((size_t)n * (N + 1), 0.0), ur((size_t)nz, 0.0);
        const int rc = almpc_set_reference(h, xr.data(), ur.data(), 0);
        if (rc != ALMPC_OK) h->designed = false;
        tr("flags + set_reference");
        return rc;
    }
    // LTV: the QP variable is v = u - ubar, so ubar takes the place of the input reference (bounds umin - ubar <= v, u = v + ubar);
    // the gradient is the explicit vector q_i (F'_i = V_i = 0: the step kernels add nothing for e0)
    {
        const int rc = almpc_set_reference(h, ltv->xbar, ltv->ubar, 1);
        if (rc != ALMPC_OK) { h->designed = false; return rc; }
    }
    h->designed = false;  // until the explicit gradient below is in place
    hipLaunchKernelGGL(k_fs_scale, dim3(256), dim3(256), 0, h->stream, h->batch, nz, nzs, h->bQ, (long)nz, h->bD, h->dFS);
    hipLaunchKernelGGL(k_neg_gm, dim3(1, (unsigned)h->batch), dim3(256), 0, h->stream, nz, nzs, 1, nz, h->bG, h->dFS, h->dV0S,
                       (long)nz * nzs, (long)nz);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->ltv = true;
    h->designed = true;
    return ALMPC_OK;
}
}  // namespace

namespace {
// almpc_relin_fnn_setup on an ALMPC_FLAG_STRUCTURED handle: network, references and weights on the device, the per-instance model and
// terminal-weight slots, and the stage-wise solver's per-instance set-up (records per stage: the terminal weight is the caller's P).
int relin_setup_structured(almpc_handle* h, int H, int L, int activation, const double* W_in, const double* W_h, const double* b_h,
                           const double* W_out, const double* xref, const double* uref, const double* Q, const double* R, const double* S,
                           const double* P, const double* umin, const double* umax) {
    const int n = h->n, m = h->m, N = h->N, nz = h->nz;
    const size_t b = (size_t)h->batch, nin = (size_t)n + m;
    if ((2 * (size_t)H + 2 * (size_t)H * nin + nin) * sizeof(double) > 160 * 1024)
        return fail(h, ALMPC_ERR_UNSUPPORTED, "relin_fnn_setup: the network's forward-mode Jacobian must fit the 160 KB of LDS");
    for (int i = 0; i < m; ++i)
        if (!(umin[i] <= umax[i])) return fail(h, ALMPC_ERR_INVALID, "relin_fnn_setup: umin > umax");
    h->designed = false;
    h->sqp.ready = h->sqp.started = false;
    h->relin.ready = false;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    auto sym = [](const double* M, int k) {
        hm::mat o((size_t)k * k, 0.0);
        if (M)
            for (int j = 0; j < k; ++j)
                for (int i = 0; i < k; ++i) o[(size_t)j * k + i] = 0.5 * (M[(size_t)j * k + i] + M[(size_t)i * k + j]);
        return o;
    };
    const hm::mat Qm = sym(Q, n), Rm = sym(R, m), Sm = sym(S, m), Pm = sym(P, n);
    const bool useS = Rm[0] != 0.0 && S && Sm[0] != 0.0;
    if (!sdual_shape_ok(n, m, N, useS))
        return fail(h, ALMPC_ERR_UNSUPPORTED, "relin_fnn_setup (structured): n (+ m with an input-rate weight) <= 48, m <= 16 and (N + 1)(n + m) <= 4096");
    almpc_handle::Relin& q = h->relin;
    for (void* p_ : {(void*)q.W_in, (void*)q.W_h, (void*)q.b_h, (void*)q.W_out, (void*)q.ulin, (void*)q.Q, (void*)q.R, (void*)q.S, (void*)q.gS})
        if (p_) (void)hipFree(p_);
    q.W_in = q.W_h = q.b_h = q.W_out = q.ulin = q.Q = q.R = q.S = q.gS = nullptr;
    auto up = [&](double** d, const double* src, size_t cnt) -> hipError_t {
        hipError_t e = dalloc(d, cnt ? cnt : 1);
        if (e == hipSuccess && src && cnt) e = hipMemcpy(*d, src, cnt * sizeof(double), hipMemcpyHostToDevice);
        return e;
    };
    std::vector<double> xr((size_t)n * (N + 1), 0.0), ur((size_t)nz, 0.0);
    if (xref) xr.assign(xref, xref + xr.size());
    if (uref) ur.assign(uref, uref + ur.size());
    std::vector<double> ul(b * m);
    for (size_t i = 0; i < b; ++i)
        for (int a = 0; a < m; ++a) ul[i * m + a] = ur[a];  // every instance linearises at the first input reference
    HIP_TRY(h, up(&q.W_in, W_in, (size_t)H * nin)); HIP_TRY(h, up(&q.W_h, W_h, (size_t)L * H * H)); HIP_TRY(h, up(&q.b_h, b_h, (size_t)L * H));
    HIP_TRY(h, up(&q.W_out, W_out, (size_t)n * H)); HIP_TRY(h, up(&q.ulin, ul.data(), ul.size()));
    if (!h->bA) HIP_TRY(h, dalloc(&h->bA, b * n * n));
    if (!h->bB) HIP_TRY(h, dalloc(&h->bB, b * n * m));
    if (!h->bP) HIP_TRY(h, dalloc(&h->bP, b * n * n));
    HIP_TRY(h, hipMemset(h->bA, 0, b * n * n * sizeof(double)));
    HIP_TRY(h, hipMemset(h->bB, 0, b * n * m * sizeof(double)));
    HIP_TRY(h, hipMemcpy(h->bP, Pm.data(), Pm.size() * sizeof(double), hipMemcpyHostToDevice));
    h->batched = true; h->ltv = false; h->r_batched_P = true; h->rP_stride = 0;
    h->sd.ready = false;
    {
        const int rc_ = sdual_setup_batched(h, Qm, Rm, useS ? &Sm : nullptr, false, h->boxmin.empty() ? nullptr : h->boxmin.data(),
                                            h->boxmax.empty() ? nullptr : h->boxmax.data(), h->terminal_eq != 0);
        if (rc_ != ALMPC_OK) return rc_;
    }
    if (riccati_shape_ok(h)) { const int rc_ = riccati_weights(h, Qm, Rm, nullptr); if (rc_ != ALMPC_OK) return rc_; }
    HIP_TRY(h, hipMemcpy(h->dUmin, umin, m * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->dUmax, umax, m * sizeof(double), hipMemcpyHostToDevice));
    h->P = Pm; h->H.clear(); h->F.clear(); h->d.clear();
    h->hS = Sm; h->useS = useS ? 1 : 0;
    h->has_box = h->boxmin.empty() ? 0 : 1;
    for (auto& e : q.ev)
        if (!e) HIP_TRY(h, hipEventCreate(&e));
    q.H = H; q.L = L; q.act = activation; q.have_prev = false; q.useR = Rm[0] != 0.0; q.useS = useS;
    if (!q.u0) HIP_TRY(h, dalloc(&q.u0, b * m));
    if (!q.xnext) HIP_TRY(h, dalloc(&q.xnext, b * n));
    h->r_has_step = false;
    h->designed = true;
    const int rc_ref = almpc_set_reference(h, xr.data(), ur.data(), 0);   // (also the input-rate terms of a horizon-varying u_ref: sdual_update_base)
    if (rc_ref != ALMPC_OK) { h->designed = false; return rc_ref; }
    q.ready = true;
    return ALMPC_OK;
}
}  // namespace

extern "C" {

int almpc_design_batched(almpc_handle* h, const double* A_batch, const double* B_batch, const double* Q, const double* R,
                         const double* S, const double* P, int P_per_instance, const double* umin, const double* umax,
                         double rho, double sigma) {
    return design_batched_common(h, A_batch, B_batch, Q, R, S, P, P_per_instance, umin, umax, rho, sigma, nullptr);
}

// Time-varying models: see include/almpc.h.  The host prepares ebar = xbar - x_ref (stages 1..N), the input part of the
// gradient 2 Rbar (ubar - u_ref) + 2 D'Sbar D ubar, and hands stage 0's (A, B) to the time-invariant slots (they only feed
// the rollout outputs x / e_x, which are not defined for an LTV design).
int almpc_design_ltv(almpc_handle* h, const double* A_all, const double* B_all, const double* c_all, const double* xbar,
                     const double* ubar, const double* xref, const double* uref, const double* Q, const double* R, const double* S,
                     const double* P, int P_per_instance, const double* umin, const double* umax, double rho, double sigma) {
    if (!h) return ALMPC_ERR_INVALID;
    drop_lazy_redo(h);
    if (!A_all || !B_all || !xbar || !ubar || !Q || !R || !P || !umin || !umax)
        return fail(h, ALMPC_ERR_INVALID, "design_ltv: null pointer (P must be given: there is no single model to take a DARE of)");
    const int n = h->n, m = h->m, N = h->N, nz = h->nz;
    const size_t b = (size_t)h->batch;
    if (!ltv_supported(h))
        return fail(h, ALMPC_ERR_UNSUPPORTED, "design_ltv: nz <= 128, or nz^2 + 3 n nz doubles must fit the 160 KB of LDS");
    std::vector<double> A0(b * n * n), B0(b * n * m), ebar(b * N * n), qadd(b * nz, 0.0);
    for (size_t i = 0; i < b; ++i) {
        std::copy(A_all + i * N * n * n, A_all + i * N * n * n + (size_t)n * n, A0.begin() + i * n * n);
        std::copy(B_all + i * N * n * m, B_all + i * N * n * m + (size_t)n * m, B0.begin() + i * n * m);
        for (int k = 0; k < N; ++k)
            for (int j = 0; j < n; ++j)
                ebar[(i * N + k) * n + j] = xbar[i * (size_t)(N + 1) * n + (size_t)(k + 1) * n + j] - (xref ? xref[(size_t)(k + 1) * n + j] : 0.0);
    }
    const bool useR = R[0] != 0.0, useS = useR && S && S[0] != 0.0;
    for (size_t i = 0; i < b; ++i) {
        const double* ub = ubar + i * nz;
        double* qa = qadd.data() + i * nz;
        if (useR)
            for (int k = 0; k < N; ++k)
                for (int a = 0; a < m; ++a) {
                    double sr = 0.0;
                    for (int c2 = 0; c2 < m; ++c2) sr += 0.5 * (R[(size_t)c2 * m + a] + R[(size_t)a * m + c2]) * (ub[k * m + c2] - (uref ? uref[k * m + c2] : 0.0));
                    qa[k * m + a] += 2.0 * sr;
                }
        if (useS)  // the input-rate cost is on u itself (src/sub/design_mpc.jl:423-446): 2 D'Sbar D ubar
            for (int k = 0; k + 1 < N; ++k)
                for (int a = 0; a < m; ++a) {
                    double sd = 0.0;
                    for (int c2 = 0; c2 < m; ++c2) sd += 0.5 * (S[(size_t)c2 * m + a] + S[(size_t)a * m + c2]) * (ub[k * m + c2] - ub[(k + 1) * m + c2]);
                    qa[k * m + a] += 2.0 * sd;
                    qa[(k + 1) * m + a] -= 2.0 * sd;
                }
    }
    LtvInputs in{A_all, B_all, c_all, ebar.data(), qadd.data(), ubar, xbar};
    return design_batched_common(h, A0.data(), B0.data(), Q, R, S, P, P_per_instance, umin, umax, rho, sigma, &in);
}

int almpc_get_gradient_instance(almpc_handle* h, int instance, double* q) {
    if (!h || !q) return ALMPC_ERR_INVALID;
    if (!h->designed || !h->ltv) return fail(h, ALMPC_ERR_NOT_DESIGNED, "get_gradient_instance needs almpc_design_ltv");
    if (instance < 0 || instance >= h->batch) return fail(h, ALMPC_ERR_INVALID, "get_gradient_instance: instance out of range");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpy(q, h->bQ + (size_t)instance * h->nz, (size_t)h->nz * sizeof(double), hipMemcpyDeviceToHost));
    return ALMPC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Per-step re-linearisation of a black-box Fnn model, resident on the device (BASELINE configs[3]; include/almpc.h).
int almpc_relin_fnn_setup(almpc_handle* h, int H, int L, int activation, const double* W_in, const double* W_h, const double* b_h,
                          const double* W_out, const double* xref, const double* uref, const double* Q, const double* R,
                          const double* S, const double* P, const double* umin, const double* umax, double rho, double sigma) {
    if (!h) return ALMPC_ERR_INVALID;
    drop_lazy_redo(h);
    if (H < 1 || L < 0 || !W_in || !W_out || (L > 0 && (!W_h || !b_h)) || !Q || !R || !P || !umin || !umax)
        return fail(h, ALMPC_ERR_INVALID, "relin_fnn_setup: null pointer or bad network shape (P must be given: the terminal weight "
                                          "comes from the linearisation at the last reference, src/sub/design_mpc.jl:312-327)");
    if (activation < 0 || activation > 4) return fail(h, ALMPC_ERR_UNSUPPORTED, "relin_fnn_setup: activation must be 0..4");
    if (h->structured) return relin_setup_structured(h, H, L, activation, W_in, W_h, b_h, W_out, xref, uref, Q, R, S, P, umin, umax);
    if (!(rho > 0.0) || !(sigma >= 0.0)) return fail(h, ALMPC_ERR_INVALID, "relin_fnn_setup: rho must be > 0 and sigma >= 0");
    const int n = h->n, m = h->m, N = h->N, nz = h->nz;
    const size_t b = (size_t)h->batch, nin = (size_t)n + m;
    if ((2 * (size_t)H + 2 * (size_t)H * nin + nin) * sizeof(double) > 160 * 1024)
        return fail(h, ALMPC_ERR_UNSUPPORTED, "relin_fnn_setup: the network's forward-mode Jacobian must fit the 160 KB of LDS");
    for (int i = 0; i < m; ++i)
        if (!(umin[i] <= umax[i])) return fail(h, ALMPC_ERR_INVALID, "relin_fnn_setup: umin > umax");
    h->designed = false;
    h->sqp.ready = h->sqp.started = false;
    h->relin.ready = false;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    auto sym = [](const double* M, int k) {
        hm::mat o((size_t)k * k, 0.0);
        if (M)
            for (int j = 0; j < k; ++j)
                for (int i = 0; i < k; ++i) o[(size_t)j * k + i] = 0.5 * (M[(size_t)j * k + i] + M[(size_t)i * k + j]);
        return o;
    };
    const hm::mat Qm = sym(Q, n), Rm = sym(R, m), Sm = sym(S, m), Pm = sym(P, n);
    { const int rc_ = ensure_batched_alloc(h); if (rc_ != ALMPC_OK) return rc_; }
    almpc_handle::Relin& q = h->relin;
    for (void* p_ : {(void*)q.W_in, (void*)q.W_h, (void*)q.b_h, (void*)q.W_out, (void*)q.ulin, (void*)q.Q, (void*)q.R, (void*)q.S, (void*)q.gS})
        if (p_) (void)hipFree(p_);
    q.W_in = q.W_h = q.b_h = q.W_out = q.ulin = q.Q = q.R = q.S = q.gS = nullptr;
    auto up = [&](double** d, const double* src, size_t cnt) -> hipError_t {
        hipError_t e = dalloc(d, cnt ? cnt : 1);
        if (e == hipSuccess && src && cnt) e = hipMemcpy(*d, src, cnt * sizeof(double), hipMemcpyHostToDevice);
        return e;
    };
    std::vector<double> xr((size_t)n * (N + 1), 0.0), ur((size_t)nz, 0.0);
    if (xref) xr.assign(xref, xref + xr.size());
    if (uref) ur.assign(uref, uref + ur.size());
    q.useR = Rm[0] != 0.0; q.useS = q.useR && Sm[0] != 0.0;
    // unscaled input-rate gradient of the shared reference: 2 D'Sbar D u_ref (the rate cost is on u itself, src/sub/design_mpc.jl:423-446)
    std::vector<double> gS((size_t)nz, 0.0);
    if (q.useS)
        for (int i = 0; i + 1 < N; ++i)
            for (int a = 0; a < m; ++a) {
                double sd = 0.0;
                for (int c2 = 0; c2 < m; ++c2) sd += Sm[(size_t)c2 * m + a] * (ur[i * m + c2] - ur[(i + 1) * m + c2]);
                gS[i * m + a] += 2.0 * sd;
                gS[(i + 1) * m + a] -= 2.0 * sd;
            }
    std::vector<double> ul(b * m);
    for (size_t i = 0; i < b; ++i)
        for (int a = 0; a < m; ++a) ul[i * m + a] = ur[a];  // every instance linearises at the first input reference
    HIP_TRY(h, up(&q.W_in, W_in, (size_t)H * nin)); HIP_TRY(h, up(&q.W_h, W_h, (size_t)L * H * H)); HIP_TRY(h, up(&q.b_h, b_h, (size_t)L * H));
    HIP_TRY(h, up(&q.W_out, W_out, (size_t)n * H)); HIP_TRY(h, up(&q.ulin, ul.data(), ul.size()));
    HIP_TRY(h, up(&q.Q, Qm.data(), Qm.size())); HIP_TRY(h, up(&q.R, Rm.data(), Rm.size())); HIP_TRY(h, up(&q.S, Sm.data(), Sm.size()));
    HIP_TRY(h, up(&q.gS, gS.data(), gS.size()));
    HIP_TRY(h, hipMemcpy(h->bP, Pm.data(), Pm.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->dUmin, umin, m * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->dUmax, umax, m * sizeof(double), hipMemcpyHostToDevice));
    // shared references; the scaled input-rate gradient fS_i = d_i .* gS and v0S_i = -G_i fS_i are per instance (re-made every step)
    for (double** pp_ : {&h->dXref, &h->dUref, &h->dFS, &h->dV0S})
        if (*pp_) { (void)hipFree(*pp_); *pp_ = nullptr; }
    h->ref_cap[0] = h->ref_cap[1] = h->ref_cap[2] = 0;
    HIP_TRY(h, up(&h->dXref, xr.data(), xr.size())); HIP_TRY(h, up(&h->dUref, ur.data(), ur.size()));
    HIP_TRY(h, dalloc(&h->dFS, b * nz)); HIP_TRY(h, dalloc(&h->dV0S, b * nz));
    HIP_TRY(h, hipMemset(h->dFS, 0, b * nz * sizeof(double))); HIP_TRY(h, hipMemset(h->dV0S, 0, b * nz * sizeof(double)));
    h->xref_stride = 0; h->uref_stride = 0; h->fS_stride = nz;
    for (auto& e : q.ev)
        if (!e) HIP_TRY(h, hipEventCreate(&e));
    q.H = H; q.L = L; q.act = activation; q.have_prev = false;
    if (!q.u0) HIP_TRY(h, dalloc(&q.u0, b * m));
    if (!q.xnext) HIP_TRY(h, dalloc(&q.xnext, b * n));
    h->P = Pm; h->hS = Sm; h->useS = q.useS;
    h->rho = rho; h->sigma = sigma;
    { const int rc_ = setup_state_rows(h, h->boxmin.empty() ? nullptr : h->boxmin.data(), h->boxmax.empty() ? nullptr : h->boxmax.data(), true);
      if (rc_ != ALMPC_OK) return rc_; }
    h->batched = true; h->ltv = false;
    h->state_valid = true;
    h->sd.ready = false;
    if (h->fallback) {
        h->r_batched_P = true; h->rP_stride = 0;
        if (sdual_shape_ok(n, m, N, q.useS != 0)) {
            // (the models change with every step: the records of the instances a step leaves unsolved are computed in that step)
            const int rc_ = sdual_setup_batched(h, Qm, Rm, q.useS ? &Sm : nullptr, false,
                                                h->boxmin.empty() ? nullptr : h->boxmin.data(), h->boxmax.empty() ? nullptr : h->boxmax.data(), h->terminal_eq != 0);
            if (rc_ != ALMPC_OK && h->fallback == 1) return rc_;
            // the references were uploaded above, not through almpc_set_reference (which this pipeline refuses): the input-rate terms
            // of a horizon-varying u_ref (+-S (u_ref[k-1] - u_ref[k])) have to reach the stage-wise redo from here
            if (rc_ == ALMPC_OK) { const int rb_ = sdual_update_base(h, ur.data(), 1); if (rb_ != ALMPC_OK) return rb_; }
        } else if (h->fallback == 1 && (h->mc > 0 || q.useS))
            return fail(h, ALMPC_ERR_UNSUPPORTED, "structured fallback: state rows / input-rate weight need n + m <= 48 and (N + 1)(n + m) <= 4096");
        if (riccati_shape_ok(h) && h->mc == 0 && !q.useS) {
            const int rc_ = riccati_weights(h, Qm, Rm, nullptr);
            if (rc_ != ALMPC_OK) return rc_;
        }
    }
    q.ready = true;
    return ALMPC_OK;
}

int almpc_relin_fnn_step_async(almpc_handle* h, const almpc_opts* opts) {
    if (!h) return ALMPC_ERR_INVALID;
    almpc_handle::Relin& q = h->relin;
    if (!q.ready) return fail(h, ALMPC_ERR_NOT_DESIGNED, "relin_fnn_step before relin_fnn_setup");
    HIP_TRY(h, hipSetDevice(h->device));
    const int n = h->n, m = h->m, nz = h->nz, nzs = h->nzs;
    hipStream_t st = h->stream;
    const bool timing = (h->flags & ALMPC_FLAG_TIMING) != 0;
    if (timing) HIP_TRY(h, hipEventRecord(q.ev[0], st));
    // 1. Jacobians at (x0_i, u_ref[:,1]) straight into the handle's per-instance model slots
    FnnParams fp;
    fp.n = n; fp.m = m; fp.H = q.H; fp.L = q.L; fp.act = q.act; fp.batch = h->batch;
    fp.W_in = q.W_in; fp.W_h = q.W_h; fp.b_h = q.b_h; fp.W_out = q.W_out;
    fp.x = h->dX0; fp.u = q.ulin; fp.ppi = 1; fp.xs_group = n; fp.us_group = m;
    fp.A = h->bA; fp.B = h->bB; fp.f = nullptr;
    if (h->structured) {
        // Structured handle (m N beyond the condensed limit, round 5): Jacobians -> the stage records of every instance's own
        // unconstrained problem (k_sgains) -> the stage-wise dual active set (k_sdual), warm-started from the previous step's inputs
        // shifted by one stage when opts->warm_start is set.  No Hessian is formed: the reference's Fnn-LP delegation has no horizon
        // limit (.../fnn/mpc_modeler_implementation_fnn.jl:23-58) and neither has this route.
        HIP_TRY(h, launch_fnn_jacobian(fp, h->num_cus, st));
        if (timing) HIP_TRY(h, hipEventRecord(q.ev[1], st));
        HIP_TRY(h, launch_sgains(h, 0));
        if (timing) HIP_TRY(h, hipEventRecord(q.ev[2], st));
        h->designed = true;
        almpc_opts o2;
        almpc_default_opts(&o2);
        if (opts) o2 = *opts;
        if (!q.have_prev) o2.warm_start = 0;
        const int rc = almpc_calculate_async(h, &o2);
        if (rc != ALMPC_OK) { h->designed = false; return rc; }
        q.have_prev = true;
        if (timing) HIP_TRY(h, hipEventRecord(q.ev[3], st));
        return ALMPC_OK;
    }
    const bool fuse_jac = design_fuses_fnn(h, q.H, q.L);   // the design kernel's workgroups linearise their own instance
    if (!fuse_jac) HIP_TRY(h, launch_fnn_jacobian(fp, h->num_cus, st));
    if (timing) HIP_TRY(h, hipEventRecord(q.ev[1], st));
    // 2. the reference's QP for every (A_i, B_i): H_i, F_i, scaling, inverses, V_i; reference-dependent vectors.  A warm step
    // (opts.warm_start = 1 after a solved step) takes its working-set guess from the previous step's inputs shifted by one stage
    // instead of an ADMM phase, and the design then needs one inverse (G_i) instead of two
    const bool warm = opts && opts->warm_start && q.have_prev && (!opts || opts->polish);
    h->skip_admm = warm ? 2 : 0;
    const DesignStrides ds = batched_strides(h, false);
    {
        const hipError_t e_ = launch_batched_design(h, ds, q.useR, q.useS, q.Q, q.R, q.S, h->rho, h->sigma, fuse_jac ? &fp : nullptr);
        if (e_ != hipSuccess) { h->skip_admm = 0; return fail(h, ALMPC_ERR_HIP, std::string("relin design: ") + hipGetErrorString(e_)); }
    }
    if (h->mc > 0) HIP_TRY(h, launch_ghat_inst(h, nullptr, nullptr));
    if (q.useS) {
        hipLaunchKernelGGL(k_fs_scale, dim3(256), dim3(256), 0, st, h->batch, nz, nzs, q.gS, 0L, h->bD, h->dFS);
        hipLaunchKernelGGL(k_neg_gm, dim3(1, (unsigned)h->batch), dim3(256), 0, st, nz, nzs, 1, nz, h->bG, h->dFS, h->dV0S,
                           (long)nz * nzs, (long)nz);
        HIP_TRY(h, hipGetLastError());
    }
    if (timing) HIP_TRY(h, hipEventRecord(q.ev[2], st));
    // 3. the step itself
    h->designed = true;
    almpc_opts o2;
    almpc_default_opts(&o2);
    if (opts) o2 = *opts;
    o2.warm_start = 0;   // (the per-instance ADMM's own warm start is not what a warm step of this pipeline means)
    // an instance whose design was flagged gets ALMPC_NON_FINITE: by the finish itself when it is polish_body (input box only), else by
    // a launch of its own behind the step
    const bool fold_flag = h->mc == 0 && o2.polish != 0;
    h->flag_in_finish = fold_flag;
    const int rc = almpc_calculate_async(h, &o2);
    h->flag_in_finish = false;
    h->skip_admm = 0;
    if (rc != ALMPC_OK) { h->designed = false; return rc; }   // (no step ran on this step's designs: the handle is not left "designed")
    q.have_prev = true;
    if (!fold_flag) {
        hipLaunchKernelGGL(k_flag_to_status, dim3((h->batch + 255) / 256), dim3(256), 0, st, h->batch, h->bFlag, h->dStatus);
        HIP_TRY(h, hipGetLastError());
    }
    if (timing) HIP_TRY(h, hipEventRecord(q.ev[3], st));
    return ALMPC_OK;
}

int almpc_relin_fnn_step(almpc_handle* h, const almpc_opts* opts) {
    const int rc = almpc_relin_fnn_step_async(h, opts);
    if (rc != ALMPC_OK) return rc;
    return almpc_synchronize(h);
}

// x0 <- fnn(x0, u[:,1]) on the device: the closed loop of the black-box model itself (the plant a simulation study drives), no host
// round trip.  Follow with almpc_relin_fnn_step(_async) -- typically with opts.warm_start = 1.
int almpc_relin_fnn_advance(almpc_handle* h) {
    if (!h) return ALMPC_ERR_INVALID;
    almpc_handle::Relin& q = h->relin;
    if (!q.ready || !q.have_prev) return fail(h, ALMPC_ERR_NOT_DESIGNED, "relin_fnn_advance needs a solved almpc_relin_fnn_step");
    HIP_TRY(h, hipSetDevice(h->device));
    { const int rc_ = enqueue_gated_redo(h); if (rc_ != ALMPC_OK) return rc_; }   // (the network is driven by decided instances' inputs only)
    const int n = h->n, m = h->m;
    hipStream_t st = h->stream;
    const size_t cnt = (size_t)h->batch * m;
    hipLaunchKernelGGL(k_pack_first_input, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, h->batch, m, h->N, h->dU, q.u0);
    FnnParams fp;
    fp.n = n; fp.m = m; fp.H = q.H; fp.L = q.L; fp.act = q.act; fp.batch = h->batch;
    fp.W_in = q.W_in; fp.W_h = q.W_h; fp.b_h = q.b_h; fp.W_out = q.W_out;
    fp.x = h->dX0; fp.u = q.u0; fp.ppi = 1; fp.xs_group = n; fp.us_group = m;
    // (the Jacobians of the forward pass go to a scratch of their own: the model slots keep the last step's linearisations, which a
    // lazily deferred redo of that step still needs)
    if (!q.Ascr) { HIP_TRY(h, dalloc(&q.Ascr, (size_t)h->batch * n * n)); HIP_TRY(h, dalloc(&q.Bscr, (size_t)h->batch * n * m)); }
    fp.A = q.Ascr; fp.B = q.Bscr; fp.f = q.xnext;
    HIP_TRY(h, launch_fnn_jacobian(fp, h->num_cus, st));
    if (h->io.x0_slot >= 0) {   // a pinned x0 slot was read once more by the forward pass: free for the host only after it
        HIP_TRY(h, hipEventRecord(h->io.ev_used[h->io.x0_slot], st));
        h->io.used_pending[h->io.x0_slot] = true;
    }
    io_release_x0(h);
    HIP_TRY(h, hipMemcpyAsync(h->dX0, q.xnext, (size_t)h->batch * n * sizeof(double), hipMemcpyDeviceToDevice, st));
    return ALMPC_OK;
}

int almpc_relin_fnn_timing(almpc_handle* h, float* ms_jacobian, float* ms_design, float* ms_step) {
    if (!h) return ALMPC_ERR_INVALID;
    if (!(h->flags & ALMPC_FLAG_TIMING) || !h->relin.ready) return fail(h, ALMPC_ERR_INVALID, "relin_fnn_timing: needs ALMPC_FLAG_TIMING and a step");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipEventSynchronize(h->relin.ev[3]));
    float a = 0, d = 0, s2 = 0;
    HIP_TRY(h, hipEventElapsedTime(&a, h->relin.ev[0], h->relin.ev[1]));
    HIP_TRY(h, hipEventElapsedTime(&d, h->relin.ev[1], h->relin.ev[2]));
    HIP_TRY(h, hipEventElapsedTime(&s2, h->relin.ev[2], h->relin.ev[3]));
    if (ms_jacobian) *ms_jacobian = a;
    if (ms_design) *ms_design = d;
    if (ms_step) *ms_step = s2;
    return ALMPC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// SQP outer loop for a black-box Fnn model, resident on the device (see almpc_sqp.hip.h and include/almpc.h).
int almpc_sqp_fnn_setup(almpc_handle* h, int H, int L, int activation, const double* W_in, const double* W_h, const double* b_h,
                        const double* W_out, const double* xref, const double* uref, const double* Q, const double* R,
                        const double* S, const double* P, int P_per_instance, const double* umin, const double* umax, double rho,
                        double sigma) {
    if (!h) return ALMPC_ERR_INVALID;
    drop_lazy_redo(h);
    if (H < 1 || L < 0 || !W_in || !W_out || (L > 0 && (!W_h || !b_h)) || !Q || !R || !P || !umin || !umax)
        return fail(h, ALMPC_ERR_INVALID, "sqp_fnn_setup: null pointer or bad network shape (P must be given)");
    if (activation < 0 || activation > 4) return fail(h, ALMPC_ERR_UNSUPPORTED, "sqp_fnn_setup: activation must be 0..4");
    if (!(rho > 0.0) || !(sigma >= 0.0)) return fail(h, ALMPC_ERR_INVALID, "sqp_fnn_setup: rho must be > 0 and sigma >= 0");
    const int n = h->n, m = h->m, N = h->N, nz = h->nz, nzs = h->nzs;
    const size_t b = (size_t)h->batch, nin = (size_t)n + m;
    if (n > 64 || sqp_step_lds_doubles(n, m, N) * sizeof(double) > 160 * 1024)
        return fail(h, ALMPC_ERR_UNSUPPORTED, "sqp_fnn_setup: n <= 64 (one lane per state in the update kernel) and its stage buffers must fit LDS");
    if (h->structured) {   // a structured handle has no condensed path: every QP of the loop goes to k_riccati (any m N <= 1024)
        if (!riccati_shape_ok(h) && !sdual_shape_ok(h->n, h->m, h->N, false))
            return fail(h, ALMPC_ERR_UNSUPPORTED, "sqp_fnn_setup: the shape is outside both stage-wise QP solvers");
        h->sqp.structured_qp = 1;
    }
    const bool sq_struct = h->sqp.structured_qp != 0;
    if (!sq_struct && !ltv_supported(h))
        return fail(h, ALMPC_ERR_UNSUPPORTED, "sqp_fnn_setup: nz <= 128, or nz^2 + 3 n nz doubles must fit the 160 KB of LDS");
    if ((2 * (size_t)H + 2 * (size_t)H * nin + nin) * sizeof(double) > 160 * 1024)
        return fail(h, ALMPC_ERR_UNSUPPORTED, "sqp_fnn_setup: the network's forward-mode Jacobian must fit the 160 KB of LDS");
    for (int i = 0; i < m; ++i)
        if (!(umin[i] <= umax[i])) return fail(h, ALMPC_ERR_INVALID, "sqp_fnn_setup: umin > umax");
    h->designed = false;  // any earlier design of the handle is replaced (its reference buffers are released below)
    h->sqp.ready = h->sqp.started = false;
    h->relin.ready = false;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    auto sym = [](const double* M, int k) {
        hm::mat o((size_t)k * k, 0.0);
        if (M)
            for (int j = 0; j < k; ++j)
                for (int i = 0; i < k; ++i) o[(size_t)j * k + i] = 0.5 * (M[(size_t)j * k + i] + M[(size_t)i * k + j]);
        return o;
    };
    const hm::mat Qm = sym(Q, n), Rm = sym(R, m), Sm = sym(S, m);
    const bool p_inst = P_per_instance != 0;
    hm::mat Pall((p_inst ? b : 1) * (size_t)n * n);
    for (size_t i = 0; i < (p_inst ? b : 1); ++i) {
        const hm::mat Pm = sym(P + i * n * n, n);
        std::copy(Pm.begin(), Pm.end(), Pall.begin() + i * n * n);
    }
    if (sq_struct && !h->batched_alloc) {   // the stage-wise QP needs no condensed operand: terminal weights and the flag words only
        if (!h->bP) HIP_TRY(h, dalloc(&h->bP, b * n * n));
        if (!h->bFlag) HIP_TRY(h, dalloc(&h->bFlag, b));
    } else {
        const int rc_ = ensure_batched_alloc(h);
        if (rc_ != ALMPC_OK) return rc_;
    }
    almpc_handle::Sqp& q = h->sqp;
    const int keep_rule = q.step_rule, keep_structured = q.structured_qp;
    void* old[] = {q.W_in, q.W_h, q.b_h, q.W_out, q.A, q.B, q.c, q.fval, q.ebar, q.qadd, q.xref, q.uref, q.Q, q.R, q.S, q.bad, q.stats, q.mer, q.xback, q.uback, q.dxback, q.vback,
                   h->dXref, h->dUref, h->dFS, h->dV0S};
    for (void* p_ : old)
        if (p_) (void)hipFree(p_);
    q = almpc_handle::Sqp();
    q.step_rule = keep_rule;
    q.structured_qp = keep_structured;
    h->dXref = h->dUref = h->dFS = h->dV0S = nullptr;
    h->ref_cap[0] = h->ref_cap[1] = h->ref_cap[2] = 0;
    auto up = [&](double** d, const double* src, size_t cnt) -> hipError_t {
        hipError_t e = dalloc(d, cnt ? cnt : 1);
        if (e == hipSuccess && src && cnt) e = hipMemcpy(*d, src, cnt * sizeof(double), hipMemcpyHostToDevice);
        return e;
    };
    std::vector<double> xr((size_t)n * (N + 1), 0.0), ur((size_t)nz, 0.0);
    if (xref) xr.assign(xref, xref + xr.size());
    if (uref) ur.assign(uref, uref + ur.size());
    HIP_TRY(h, up(&q.W_in, W_in, (size_t)H * nin)); HIP_TRY(h, up(&q.W_h, W_h, (size_t)L * H * H)); HIP_TRY(h, up(&q.b_h, b_h, (size_t)L * H));
    HIP_TRY(h, up(&q.W_out, W_out, (size_t)n * H));
    HIP_TRY(h, up(&q.A, nullptr, b * N * n * n)); HIP_TRY(h, up(&q.B, nullptr, b * N * n * m)); HIP_TRY(h, up(&q.c, nullptr, b * N * n));
    HIP_TRY(h, up(&q.fval, nullptr, b * N * n)); HIP_TRY(h, up(&q.ebar, nullptr, b * N * n)); HIP_TRY(h, up(&q.qadd, nullptr, b * nz));
    HIP_TRY(h, up(&q.xref, xr.data(), xr.size())); HIP_TRY(h, up(&q.uref, ur.data(), ur.size()));
    HIP_TRY(h, up(&q.Q, Qm.data(), Qm.size())); HIP_TRY(h, up(&q.R, Rm.data(), Rm.size())); HIP_TRY(h, up(&q.S, Sm.data(), Sm.size()));
    HIP_TRY(h, dalloc(&q.bad, b));
    HIP_TRY(h, dalloc(&q.mer, 4 * b));
    HIP_TRY(h, dalloc(&q.xback, b * (size_t)n * (N + 1))); HIP_TRY(h, dalloc(&q.dxback, b * (size_t)n * (N + 1)));
    HIP_TRY(h, dalloc(&q.uback, b * nz)); HIP_TRY(h, dalloc(&q.vback, b * nz));
    {
        double pm = 0.0;
        for (double v : Pall) pm = std::max(pm, std::fabs(v));
        for (double v : Qm) pm = std::max(pm, std::fabs(v));
        q.mu = 2.0 * pm;  // the multipliers of the dynamics are ~ 2 |P e|: exact for errors up to order one.  Measured on the benchmark
                          // set: 0.2 |P| rejects good steps near the solution, 10 |P| rejects every full step of some instances
    }
    HIP_TRY(h, hipMemset(q.bad, 0, b * sizeof(int)));
    HIP_TRY(h, dalloc(&h->dXref, b * (size_t)n * (N + 1))); HIP_TRY(h, dalloc(&h->dUref, b * nz));
    HIP_TRY(h, hipMemcpy(h->bP, Pall.data(), Pall.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->dUmin, umin, m * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->dUmax, umax, m * sizeof(double), hipMemcpyHostToDevice));
    if (h->batched_alloc) {   // the condensed route (also kept ready when a condensed handle sends its QPs to k_riccati after an earlier design)
        if (!h->bQ) HIP_TRY(h, dalloc(&h->bQ, b * nz));
        HIP_TRY(h, dalloc(&h->dFS, b * nz)); HIP_TRY(h, dalloc(&h->dV0S, b * nz));
        // F_i = 0 for an LTV design (the gradient is explicit), so F'_i and V_i stay zero; stage-0 model slots are unused but read
        HIP_TRY(h, hipMemset(h->bF, 0, b * nz * n * sizeof(double)));
        HIP_TRY(h, hipMemset(h->bFs, 0, b * n * nzs * sizeof(double)));
        HIP_TRY(h, hipMemset(h->bVs, 0, b * n * nzs * sizeof(double)));
        HIP_TRY(h, hipMemset(h->bA, 0, b * n * n * sizeof(double)));
        HIP_TRY(h, hipMemset(h->bB, 0, b * n * m * sizeof(double)));
    }
    q.H = H; q.L = L; q.act = activation; q.useR = Rm[0] != 0.0; q.useS = q.useR && Sm[0] != 0.0; q.sP = p_inst ? (long)n * n : 0;
    h->P.assign(Pall.begin(), Pall.begin() + (size_t)n * n);
    h->hS = Sm;
    h->useS = q.useS;
    h->rho = rho; h->sigma = sigma;
    // state rows (almpc_set_state_box: the box of .../fnn/mpc_modeler_implementation_fnn.jl:146-153; terminal equality): one
    // constraint-space matrix per instance, rebuilt with every iteration's linearisation
    const double* bxmin = h->boxmin.empty() ? nullptr : h->boxmin.data();
    const double* bxmax = h->boxmax.empty() ? nullptr : h->boxmax.data();
    if (q.structured_qp) {   // no constraint-space matrix: the state rows are coordinates of the stage-wise trajectory (k_sdual)
        const int rc_ = setup_state_rows(h, nullptr, nullptr, true);
        if (rc_ != ALMPC_OK) return rc_;
        h->has_box = bxmin ? 1 : 0;
        h->mc = bxmin ? N * n : (h->terminal_eq ? n : 0);
    } else {
        const int rc_ = setup_state_rows(h, bxmin, bxmax, true);
        if (rc_ != ALMPC_OK) return rc_;
    }
    h->xref_stride = (long)n * (N + 1); h->uref_stride = nz; h->fS_stride = nz;
    h->designed = false;  // becomes true with the first iteration's design
    h->batched = true; h->ltv = true;
    h->r_batched_P = true; h->rP_stride = q.sP;
    h->sd.ready = false;
    if (h->fallback || q.structured_qp) {
        // stage-wise QP of an iteration: k_sgains (stage records, defects' value-function terms, cost terms) + k_sdual; behind it, for
        // an input box without S, the primal Riccati active set
        if (sdual_shape_ok(n, m, N, q.useS != 0)) {
            const int rc_ = sdual_setup_batched(h, Qm, Rm, q.useS ? &Sm : nullptr, false, bxmin, bxmax, h->terminal_eq != 0);
            if (rc_ != ALMPC_OK && (h->fallback == 1 || q.structured_qp)) return rc_;
            if (rc_ == ALMPC_OK) {
                almpc_handle::Sd& sd = h->sd;
                const size_t TP = (size_t)sdual_tp(sd.NT, sd.MC, N);
                if (sd.base_cap < b * TP) {
                    if (sd.base) { (void)hipFree(sd.base); sd.base = nullptr; }
                    HIP_TRY(h, dalloc(&sd.base, b * TP));
                    sd.base_cap = b * TP;
                }
                sd.has_base = true; sd.base_stride = (long)TP;
                if (sd.pc) { (void)hipFree(sd.pc); sd.pc = nullptr; }
                if (sd.ct) { (void)hipFree(sd.ct); sd.ct = nullptr; }
                HIP_TRY(h, dalloc(&sd.pc, b * N * sd.NT)); HIP_TRY(h, dalloc(&sd.ct, b * N * sd.NT));
                sd.sqp = true;
            }
        } else if ((q.structured_qp || h->fallback == 1) && (h->mc > 0 || q.useS))
            return fail(h, ALMPC_ERR_UNSUPPORTED, "sqp_fnn_setup: state rows / input-rate weight in the stage-wise QP need n + m <= 48 and (N + 1)(n + m) <= 4096");
        if (riccati_shape_ok(h) && h->mc == 0 && !q.useS) {
            const int rc_ = riccati_weights(h, Qm, Rm, nullptr);
            if (rc_ != ALMPC_OK) return rc_;
        } else if (!h->sd.ready && q.structured_qp)
            return fail(h, ALMPC_ERR_UNSUPPORTED, "sqp_fnn_setup: the shape is outside both stage-wise QP solvers");
    }
    q.ready = true; q.started = false;
    return ALMPC_OK;
}

int almpc_sqp_fnn_start(almpc_handle* h, const double* x0, const double* u_guess) {
    if (!h || !x0) return h ? fail(h, ALMPC_ERR_INVALID, "sqp_fnn_start: null x0") : ALMPC_ERR_INVALID;
    almpc_handle::Sqp& q = h->sqp;
    if (!q.ready) return fail(h, ALMPC_ERR_NOT_DESIGNED, "sqp_fnn_start before sqp_fnn_setup");
    HIP_TRY(h, hipSetDevice(h->device));
    const int n = h->n, m = h->m, N = h->N, nz = h->nz;
    const size_t b = (size_t)h->batch;
    hipStream_t st = h->stream;
    HIP_TRY(h, hipStreamSynchronize(st));
    io_release_x0(h);
    HIP_TRY(h, hipMemcpy(h->dX0, x0, b * n * sizeof(double), hipMemcpyHostToDevice));
    if (u_guess) {
        std::vector<double> ug(u_guess, u_guess + b * nz), lo(m), hi(m);
        HIP_TRY(h, hipMemcpy(lo.data(), h->dUmin, m * sizeof(double), hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(hi.data(), h->dUmax, m * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t t = 0; t < ug.size(); ++t) ug[t] = std::min(std::max(ug[t], lo[t % m]), hi[t % m]);
        HIP_TRY(h, hipMemcpy(h->dUref, ug.data(), ug.size() * sizeof(double), hipMemcpyHostToDevice));
    } else {  // the input reference, clipped to the box
        std::vector<double> ur(nz), lo(m), hi(m), ug(b * nz);
        HIP_TRY(h, hipMemcpy(ur.data(), q.uref, nz * sizeof(double), hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(lo.data(), h->dUmin, m * sizeof(double), hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(hi.data(), h->dUmax, m * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t t = 0; t < ug.size(); ++t) ug[t] = std::min(std::max(ur[t % nz], lo[t % m]), hi[t % m]);
        HIP_TRY(h, hipMemcpy(h->dUref, ug.data(), ug.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    FnnRolloutParams rp;
    rp.n = n; rp.m = m; rp.H = q.H; rp.L = q.L; rp.act = q.act; rp.N = N;
    rp.W_in = q.W_in; rp.W_h = q.W_h; rp.b_h = q.b_h; rp.W_out = q.W_out; rp.x0 = h->dX0; rp.ubar = h->dUref; rp.xbar = h->dXref;
    const size_t l = (2 * (size_t)q.H + n + m) * sizeof(double);
    if (l > 64 * 1024) HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(k_fnn_rollout), (size_t)(l)));
    hipLaunchKernelGGL(k_fnn_rollout, dim3((unsigned)b), dim3(256), l, st, rp);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemsetAsync(q.bad, 0, b * sizeof(int), st));
    {
        std::vector<double> d0(4 * b, 0.0);
        for (size_t i = 0; i < b; ++i) { d0[4 * i] = 1.0; d0[4 * i + 1] = std::numeric_limits<double>::infinity(); }
        HIP_TRY(h, hipMemcpyAsync(q.mer, d0.data(), d0.size() * sizeof(double), hipMemcpyHostToDevice, st));
        HIP_TRY(h, hipStreamSynchronize(st));
    }
    q.started = true;
    q.since_start = 0;
    { const char* e = getenv("ALMPC_SQP_ADMM_ALWAYS"); q.guess_from_iterate = (e && e[0] == '1') ? 0 : 1; }
    return ALMPC_OK;
}

int almpc_sqp_fnn_iterate(almpc_handle* h, int iters, double step_scale, const almpc_opts* opts, double* step_inf, double* defect_inf) {
    if (!h) return ALMPC_ERR_INVALID;
    almpc_handle::Sqp& q = h->sqp;
    if (!q.ready || !q.started) return fail(h, ALMPC_ERR_NOT_DESIGNED, "sqp_fnn_iterate needs sqp_fnn_setup and sqp_fnn_start");
    if (iters < 1 || !(step_scale > 0.0 && step_scale <= 1.0)) return fail(h, ALMPC_ERR_INVALID, "sqp_fnn_iterate: iters >= 1, 0 < step_scale <= 1");
    HIP_TRY(h, hipSetDevice(h->device));
    const int n = h->n, m = h->m, N = h->N, nz = h->nz, nzs = h->nzs;
    const size_t b = (size_t)h->batch;
    hipStream_t st = h->stream;
    if (q.stats_cap < iters) {
        if (q.stats) { HIP_TRY(h, hipStreamSynchronize(st)); (void)hipFree(q.stats); q.stats = nullptr; }
        HIP_TRY(h, dalloc(&q.stats, (size_t)2 * iters));
        q.stats_cap = iters;
    }
    HIP_TRY(h, hipMemsetAsync(q.stats, 0, (size_t)2 * iters * sizeof(unsigned long long), st));
    const DesignStrides ds = batched_strides(h, q.sP != 0);
    const size_t step_lds = sqp_step_lds_doubles(n, m, N) * sizeof(double);
    if (step_lds > 64 * 1024) HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(k_sqp_step), (size_t)(step_lds)));
    FnnParams fp;
    fp.n = n; fp.m = m; fp.H = q.H; fp.L = q.L; fp.act = q.act; fp.batch = (int)(b * N);
    fp.W_in = q.W_in; fp.W_h = q.W_h; fp.b_h = q.b_h; fp.W_out = q.W_out;
    fp.x = h->dXref; fp.u = h->dUref; fp.ppi = N; fp.xs_group = (long)n * (N + 1); fp.us_group = nz;
    fp.A = q.A; fp.B = q.B; fp.f = q.fval;
    SqpParams sp;
    sp.n = n; sp.m = m; sp.N = N; sp.nz = nz; sp.batch = h->batch; sp.useR = q.useR; sp.useS = q.useS;
    sp.xref = q.xref; sp.uref = q.uref; sp.R = q.R; sp.S = q.S; sp.umin = h->dUmin; sp.umax = h->dUmax;
    sp.xbar = h->dXref; sp.ubar = h->dUref; sp.fval = q.fval; sp.A = q.A; sp.B = q.B; sp.c = q.c; sp.ebar = q.ebar; sp.qadd = q.qadd;
    sp.v = h->dEu; sp.flag = h->bFlag; sp.status = h->dStatus; sp.bad = q.bad; sp.stats = q.stats; sp.step_scale = step_scale;
    sp.x = h->dX; sp.ex = h->dEx; sp.u = h->dU; sp.eu = h->dEu; sp.adaptive = q.step_rule; sp.mu = q.mu; sp.Q = q.Q; sp.P = h->bP; sp.sP = q.sP; sp.mer = q.mer;
    sp.xback = q.xback; sp.uback = q.uback; sp.dxback = q.dxback; sp.vback = q.vback;
    DesignLtvParams lp;
    lp.n = n; lp.m = m; lp.N = N; lp.nz = nz; lp.useR = q.useR; lp.useS = q.useS;
    lp.A = q.A; lp.B = q.B; lp.c = q.c; lp.ebar = q.ebar; lp.P = h->bP; lp.sP = q.sP; lp.Q = q.Q; lp.R = q.R; lp.S = q.S;
    lp.qadd = q.qadd; lp.H = h->bH; lp.q = h->bQ;
    // register-tile design kernel: scaling, scaled gradient and the flag reset ride along as its tail (three launches less per iteration)
    const bool ltv_scales = ltv_reg_path(h) && nz <= 128 && !q.structured_qp && !getenv("ALMPC_DBG_SPLIT_SCALE");
    if (ltv_scales) { lp.sc_d = h->bD; lp.sc_Hs = h->bHs; lp.sc_fS = h->dFS; lp.sc_flag = h->bFlag; lp.nzs = nzs; }
    // ... and k_sqp_prepare as its head (its outputs are that kernel's inputs): a fourth launch less
    const bool prep_in_design = ltv_scales && !getenv("ALMPC_DBG_SPLIT_PREPARE");
    if (prep_in_design) { lp.prep_on = 1; lp.prep = sp; }
    for (int it = 0; it < iters; ++it) {
        HIP_TRY(h, launch_fnn_jacobian(fp, h->num_cus, st));
        if (!prep_in_design) hipLaunchKernelGGL(k_sqp_prepare, dim3((unsigned)b), dim3(256), 0, st, sp);
        if (!ltv_scales) HIP_TRY(h, hipMemsetAsync(h->bFlag, 0, b * sizeof(int), st));
        if (q.structured_qp) {   // the QP in its stage-wise form for every instance; start: v = 0 (working set = the iterate's inputs on a bound)
            h->designed = true;
            // input box without S: the primal Riccati active set (an iterate of this loop has about half of its inputs on a bound: rows a
            // primal method holds at no cost, while the dual one pays a sweep per row of its start -- measured 0.56 against 3.1 ms per
            // iteration at the configs[4] shape); with state rows / S: k_sgains + k_sdual
            if (h->mc == 0 && !q.useS && h->rKst)
                HIP_TRY(h, launch_riccati(h, 3, h->dUref, opts ? opts->polish_max_iter : 0));
            else if (h->sd.ready) {
                HIP_TRY(h, launch_sgains(h, 0));
                HIP_TRY(h, launch_sdual(h, 0, h->dUref, opts ? opts->polish_max_iter : 0, false, h->nz > 48 ? 1 : 0));
            } else
                return fail(h, ALMPC_ERR_UNSUPPORTED, "sqp_fnn_iterate: no stage-wise QP solver for this design");
            q.since_start += 1;
            sp.stats = q.stats + 2 * it;
            hipLaunchKernelGGL(k_sqp_step, dim3((unsigned)b), dim3(256), step_lds, st, sp);
            HIP_TRY(h, hipGetLastError());
            continue;
        }
        HIP_TRY(h, launch_design_ltv(h, lp, st));
        h->skip_admm = (q.guess_from_iterate && q.since_start > 0) ? 1 : 0;
        // (with the scaling in the design kernel's tail fS_i exists before the inverse: v0S_i = -G_i fS_i comes out of the inverse's launch)
        const bool v0_in_inverse = ltv_scales && design_inverse_makes_v(nz);
        launch_batched_factor(h, ds, h->rho, h->sigma, st, ltv_scales, false, v0_in_inverse ? h->dFS : nullptr, h->dV0S);
        if (h->mc > 0) HIP_TRY(h, launch_ghat_inst(h, q.A, q.B));
        if (!ltv_scales) hipLaunchKernelGGL(k_fs_scale, dim3(256), dim3(256), 0, st, h->batch, nz, nzs, h->bQ, (long)nz, h->bD, h->dFS);
        if (!v0_in_inverse) hipLaunchKernelGGL(k_neg_gm, dim3(1, (unsigned)b), dim3(256), 0, st, nz, nzs, 1, nz, h->bG, h->dFS, h->dV0S, (long)nz * nzs, (long)nz);
        HIP_TRY(h, hipGetLastError());
        h->designed = true;
        const int rc = almpc_calculate_async(h, opts);
        h->skip_admm = 0;
        if (rc != ALMPC_OK) return rc;
        // structured fallback: an instance whose condensed Hessian came out indefinite to working precision (open-loop unstable
        // linearisation over the horizon) or whose QP was left unsolved gets this iteration's QP solved in its stage-wise form
        if (h->fallback) {
            // input box without S: the primal Riccati active set alone -- the solver this loop's stage-wise QP route uses for such problems
            // (an iterate holds half of its inputs on bounds; a saturated unstable linearisation is where the dual method has no
            // certificate) --, ONE idle launch per iteration instead of three; with state rows / S: k_sgains + k_sdual
            const bool primal_only = h->mc == 0 && !q.useS && h->rKst && !getenv("ALMPC_SQP_REDO_DUAL_FIRST");
            if (h->sd.ready && !primal_only) { HIP_TRY(h, launch_sgains(h, 2)); HIP_TRY(h, launch_sdual(h, 2, nullptr, 0, true)); }
            if (h->mc == 0 && !q.useS && h->rKst) HIP_TRY(h, launch_riccati(h, 2, nullptr, 0));
        }
        q.since_start += 1;
        sp.stats = q.stats + 2 * it;
        hipLaunchKernelGGL(k_sqp_step, dim3((unsigned)b), dim3(256), step_lds, st, sp);
        HIP_TRY(h, hipGetLastError());
    }
    std::vector<unsigned long long> stats((size_t)2 * iters);
    std::vector<int> bad(b);
    HIP_TRY(h, hipMemcpyAsync(stats.data(), q.stats, stats.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(bad.data(), q.bad, b * sizeof(int), hipMemcpyDeviceToHost, st));
    { const int rc = almpc_synchronize(h); if (rc != ALMPC_OK) return rc; }
    for (int it = 0; it < iters; ++it) {
        double a, c2;
        std::memcpy(&a, &stats[2 * it], 8); std::memcpy(&c2, &stats[2 * it + 1], 8);
        if (step_inf) step_inf[it] = a;
        if (defect_inf) defect_inf[it] = c2;
    }
    for (size_t i = 0; i < b; ++i)
        if (bad[i])
            return fail(h, ALMPC_ERR_NUMERIC, "sqp_fnn_iterate: instance " + std::to_string(i) +
                        ": an iteration was skipped (condensed Hessian not positive definite to working precision, a non-finite QP "
                        "solution, or -- with state rows -- an infeasible QP); its iterate is the last good one, the other instances are unaffected");
    return ALMPC_OK;
}

int almpc_sqp_fnn_set_structured(almpc_handle* h, int on) {
    if (!h) return ALMPC_ERR_INVALID;
    if (on && !riccati_shape_ok(h) && !sdual_shape_ok(h->n, h->m, h->N, false))
        return fail(h, ALMPC_ERR_UNSUPPORTED, "sqp_fnn_set_structured: the shape is outside both stage-wise QP solvers");
    h->sqp.structured_qp = on ? 1 : 0;
    h->designed = false;   // takes effect at the next almpc_sqp_fnn_setup
    h->sqp.ready = h->sqp.started = false;
    return ALMPC_OK;
}

int almpc_sqp_fnn_set_step_rule(almpc_handle* h, int rule) {
    if (!h) return ALMPC_ERR_INVALID;
    if (rule != 0 && rule != 1) return fail(h, ALMPC_ERR_INVALID, "sqp_fnn_set_step_rule: 0 (fixed step) or 1 (merit-function safeguard)");
    h->sqp.step_rule = rule;
    return ALMPC_OK;
}

int almpc_sqp_fnn_skipped(almpc_handle* h, int32_t* skipped) {
    if (!h || !skipped) return ALMPC_ERR_INVALID;
    if (!h->sqp.ready) return fail(h, ALMPC_ERR_NOT_DESIGNED, "sqp_fnn_skipped before sqp_fnn_setup");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(skipped, h->sqp.bad, (size_t)h->batch * sizeof(int), hipMemcpyDeviceToHost));
    return ALMPC_OK;
}

int almpc_get_design_instance(almpc_handle* h, int instance, double* H, double* F, double* d) {
    if (!h) return ALMPC_ERR_INVALID;
    if (!h->designed || !h->batched) return fail(h, ALMPC_ERR_NOT_DESIGNED, "get_design_instance needs almpc_design_batched");
    if (instance < 0 || instance >= h->batch) return fail(h, ALMPC_ERR_INVALID, "get_design_instance: instance out of range");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t nz = h->nz, n = h->n, i = (size_t)instance;
    if (H) HIP_TRY(h, hipMemcpy(H, h->bH + i * nz * nz, nz * nz * sizeof(double), hipMemcpyDeviceToHost));
    if (F) HIP_TRY(h, hipMemcpy(F, h->bF + i * nz * n, nz * n * sizeof(double), hipMemcpyDeviceToHost));
    if (d) HIP_TRY(h, hipMemcpy(d, h->bD + i * h->nzs, nz * sizeof(double), hipMemcpyDeviceToHost));
    return ALMPC_OK;
}

int almpc_set_reference(almpc_handle* h, const double* xref, const double* uref, int per_instance) {
    if (!h) return ALMPC_ERR_INVALID;
    if (!h->designed) return fail(h, ALMPC_ERR_NOT_DESIGNED, "set_reference before design");
    if (h->ltv) return fail(h, ALMPC_ERR_INVALID, "set_reference: the references of an LTV design are arguments of almpc_design_ltv");
    if (h->relin.ready)
        return fail(h, ALMPC_ERR_INVALID, "set_reference: the references of the re-linearisation pipeline are arguments of almpc_relin_fnn_setup");
    if (!xref || !uref) return fail(h, ALMPC_ERR_INVALID, "set_reference: null pointer");
    HIP_TRY(h, hipSetDevice(h->device));
    const int n = h->n, m = h->m, N = h->N, nz = h->nz;
    const size_t cnt = per_instance ? (size_t)h->batch : 1;
    const size_t xs = (size_t)n * (N + 1), us = (size_t)nz;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->structured) {
        h->designed = false;
        if (h->dXref) { (void)hipFree(h->dXref); h->dXref = nullptr; }
        if (h->dUref) { (void)hipFree(h->dUref); h->dUref = nullptr; }
        h->ref_cap[0] = h->ref_cap[1] = h->ref_cap[2] = 0;
        HIP_TRY(h, dalloc(&h->dXref, cnt * xs));
        HIP_TRY(h, dalloc(&h->dUref, cnt * us));
        HIP_TRY(h, hipMemcpy(h->dXref, xref, cnt * xs * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->dUref, uref, cnt * us * sizeof(double), hipMemcpyHostToDevice));
        h->xref_stride = per_instance ? (long)xs : 0;
        h->uref_stride = per_instance ? (long)us : 0;
        { const int rc_ = sdual_update_base(h, uref, cnt); if (rc_ != ALMPC_OK) return rc_; }
        h->designed = true;
        return ALMPC_OK;
    }
    h->sqp.ready = h->sqp.started = false;  // the SQP iterate lived in the reference buffers released here
    h->relin.ready = false;                 // ... and so did the re-linearisation pipeline's references
    h->designed = false;                    // the reference buffers are replaced below: designed again on success only
    const size_t fcnt = h->batched ? (size_t)h->batch : cnt;  // per-instance models: fS depends on d_i
    // (a re-design or a new reference of the same shape keeps the four buffers: hipFree + hipMalloc cost ~0.1 ms each)
    if (!(h->dXref && h->dUref && h->dFS && h->dV0S && h->ref_cap[0] == cnt * xs && h->ref_cap[1] == cnt * us && h->ref_cap[2] == fcnt * us)) {
        if (h->dXref) { (void)hipFree(h->dXref); h->dXref = nullptr; }
        if (h->dUref) { (void)hipFree(h->dUref); h->dUref = nullptr; }
        if (h->dFS) { (void)hipFree(h->dFS); h->dFS = nullptr; }
        if (h->dV0S) { (void)hipFree(h->dV0S); h->dV0S = nullptr; }
        h->ref_cap[0] = h->ref_cap[1] = h->ref_cap[2] = 0;
        HIP_TRY(h, dalloc(&h->dXref, cnt * xs));
        HIP_TRY(h, dalloc(&h->dUref, cnt * us));
        HIP_TRY(h, dalloc(&h->dFS, fcnt * us));
        HIP_TRY(h, dalloc(&h->dV0S, fcnt * us));
        h->ref_cap[0] = cnt * xs; h->ref_cap[1] = cnt * us; h->ref_cap[2] = fcnt * us;
    }
    HIP_TRY(h, hipMemcpy(h->dXref, xref, cnt * xs * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->dUref, uref, cnt * us * sizeof(double), hipMemcpyHostToDevice));
    // fS = d .* (2 D'Sbar D u_ref): the input-rate cost is on u, not e_u (src/sub/design_mpc.jl:423-446)
    std::vector<double> fS(cnt * us, 0.0);
    if (h->useS) {  // S counts only together with R (the reference's branch rule, src/sub/design_mpc.jl:423-466), as in the design kernels
        for (size_t c = 0; c < cnt; ++c) {
            const double* ur = uref + c * us;
            double* f = fS.data() + c * us;
            for (int i = 0; i + 1 < N; ++i)
                for (int a = 0; a < m; ++a) {
                    double sd = 0.0;  // (S (u_i - u_{i+1}))_a
                    for (int b2 = 0; b2 < m; ++b2) sd += h->hS[(size_t)b2 * m + a] * (ur[i * m + b2] - ur[(i + 1) * m + b2]);
                    f[i * m + a] += 2.0 * sd;
                    f[(i + 1) * m + a] -= 2.0 * sd;
                }
            if (!h->batched)
                for (int r = 0; r < nz; ++r) f[r] *= h->d[r];
        }
    }
    if (h->batched && !h->useS) {   // no input-rate weight: fS_i = 0 and v0S_i = -G_i fS_i = 0 (no pass over the 0.5 GB of G_i)
        HIP_TRY(h, hipMemsetAsync(h->dFS, 0, fcnt * us * sizeof(double), h->stream));
        HIP_TRY(h, hipMemsetAsync(h->dV0S, 0, fcnt * us * sizeof(double), h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    } else if (h->batched) {  // fS_i = d_i .* g and v0S_i = -G_i fS_i on the device, one vector per instance
        double* dGs = nullptr;
        HIP_TRY(h, dalloc(&dGs, cnt * us));
        hipError_t e = hipMemcpy(dGs, fS.data(), cnt * us * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_fs_scale, dim3(256), dim3(256), 0, h->stream, h->batch, nz, h->nzs, dGs, per_instance ? (long)us : 0L,
                               h->bD, h->dFS);
            hipLaunchKernelGGL(k_neg_gm, dim3(1, (unsigned)h->batch), dim3(256), 0, h->stream, nz, h->nzs, 1, (int)us, h->bG, h->dFS,
                               h->dV0S, (long)nz * h->nzs, (long)us);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        }
        (void)hipFree(dGs);
        if (e != hipSuccess) return fail(h, ALMPC_ERR_HIP, std::string("set_reference (batched): ") + hipGetErrorString(e));
    } else {
    HIP_TRY(h, hipMemcpy(h->dFS, fS.data(), cnt * us * sizeof(double), hipMemcpyHostToDevice));
    // v0S = -G fS: constant part of the polish's unconstrained minimiser (k_admm adds the part linear in e0)
    {
        const size_t blocks = (cnt * us + 255) / 256;
        hipLaunchKernelGGL(k_neg_gm, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, h->stream, nz, h->nzs,
                           (int)cnt, (int)us, h->dG, h->dFS, h->dV0S, 0L, 0L);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    }
    h->xref_stride = per_instance ? (long)xs : 0;
    h->uref_stride = per_instance ? (long)us : 0;
    h->fS_stride = (per_instance || h->batched) ? (long)us : 0;
    { const int rc_ = sdual_update_base(h, uref, cnt); if (rc_ != ALMPC_OK) return rc_; }
    { const int rc_ = build_s0_basis(h); if (rc_ != ALMPC_OK) return rc_; }   // (v0S has changed)
    h->designed = true;
    return ALMPC_OK;
}

int almpc_update_initialization(almpc_handle* h, const double* x0) {
    if (!h || !x0) return h ? fail(h, ALMPC_ERR_INVALID, "update_initialization: null x0") : ALMPC_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    io_release_x0(h);
    HIP_TRY(h, hipMemcpyAsync(h->dX0, x0, (size_t)h->batch * h->n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return ALMPC_OK;
}

int almpc_update_initialization_device(almpc_handle* h, const double* d_x0) {
    if (!h || !d_x0) return h ? fail(h, ALMPC_ERR_INVALID, "update_initialization_device: null x0") : ALMPC_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    io_release_x0(h);
    HIP_TRY(h, hipMemcpyAsync(h->dX0, d_x0, (size_t)h->batch * h->n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return ALMPC_OK;
}

int almpc_calculate_async(almpc_handle* h, const almpc_opts* user) {
    if (!h) return ALMPC_ERR_INVALID;
    if (!h->designed) return fail(h, ALMPC_ERR_NOT_DESIGNED, "calculate before design");
    almpc_opts o;
    almpc_default_opts(&o);
    if (user) o = *user;
    if (o.rho == 0.0) o.rho = h->rho;
    if (o.sigma == 0.0) o.sigma = h->sigma;
    if (o.rho != h->rho || o.sigma != h->sigma)
        return fail(h, ALMPC_ERR_INVALID, "calculate: rho/sigma differ from the design values (the shared KKT inverse is built for them)");
    if (o.max_iter < 1 || o.check_every < 1 || !(o.alpha > 0.0 && o.alpha < 2.0) || !(o.eps_abs >= 0.0) || !(o.eps_rel >= 0.0))
        return fail(h, ALMPC_ERR_INVALID, "calculate: bad options");
    // ALMPC_OPT_NO_WARM_STATE: shared-model steps with the polish on (the polish needs only the signs of y); ignored elsewhere
    const bool keep_state = !((o.reserved[0] & ALMPC_OPT_NO_WARM_STATE) && !h->batched && o.polish && h->mc == 0);
    if (o.warm_start && !h->state_valid)
        return fail(h, ALMPC_ERR_INVALID, "calculate: warm_start = 1, but the previous step ran with ALMPC_OPT_NO_WARM_STATE (no ADMM state was kept)");
    // default redo of unsolved instances on the shared-model / input-box-only path: lazily, at the next host sync (see hUnsolved)
    // (every condensed path that is not the SQP loop ends in a finish that does the counting -- polish_body for an input box,
    // polish_gen_body with state rows --: shared model, per-instance models, re-linearisation pipeline.  Infeasible instances keep
    // their verdict and are not counted; what is counted is rare -- a handful of edge-of-feasibility instances in 4096 -- and a
    // stage-wise solve of one of them takes about a millisecond, which an eager redo would put behind every step)
    const bool lazy_redo = h->fallback == 2 && !h->structured && !h->ltv && o.polish != 0 &&
                           (h->sd.ready || (h->mc == 0 && h->rKst)) && !getenv("ALMPC_EAGER_REDO");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->io.big_copy_pending) {   // an asynchronous read-back straight from the result buffers: this step overwrites them
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->io.ev_big, 0));
        h->io.big_copy_pending = false;
    }
    auto ensure_unsolved_word = [&]() -> hipError_t {   // the host-visible counter of the lazy redo
        if (h->hUnsolved) return hipSuccess;
        hipError_t e_ = hipHostMalloc(reinterpret_cast<void**>(&h->hUnsolved), sizeof(int), hipHostMallocMapped);
        if (e_ != hipSuccess) { h->hUnsolved = nullptr; return e_; }
        *h->hUnsolved = 0; h->unsolved_seen = 0;
        e_ = dalloc(&h->dRedoGate, 1);
        if (e_ != hipSuccess) return e_;
        e_ = hipMemset(h->dRedoGate, 0, sizeof(int));   // (step numbers start at 1)
        if (e_ != hipSuccess) return e_;
        return hipHostGetDevicePointer(reinterpret_cast<void**>(&h->dUnsolved), h->hUnsolved, 0);
    };
    h->step_serial += 1;
    h->sd.start_ws_fresh = false;
    auto io_step_done = [&]() -> int {   // the x0 slot of an asynchronous update is free again once this step has finished
        h->state_valid = keep_state;     // (recorded only here: every launch of the step went out)
        if (h->io.x0_slot >= 0) {
            HIP_TRY(h, hipEventRecord(h->io.ev_used[h->io.x0_slot], h->stream));
            h->io.used_pending[h->io.x0_slot] = true;
        }
        return ALMPC_OK;
    };
    if (h->structured) {   // the Riccati active-set solve is the whole step (polish_max_iter caps its working-set changes)
        const double* guess = nullptr;
        if (h->guess_ready) { guess = h->rGuess; h->guess_ready = false; }   // almpc_set_start_from: consumed by this step
        else if (o.warm_start && h->r_has_step) {   // receding horizon: the previous step's inputs shifted by one stage (the last stage repeated)
            if (!h->rGuess) HIP_TRY(h, dalloc(&h->rGuess, (size_t)h->batch * h->nz));
            const long cnt = (long)h->batch * h->nz;
            hipLaunchKernelGGL(k_guess_from_inputs, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, h->batch, h->m, h->N, h->N, 1,
                               (const double*)h->dU, (const double*)h->dUref, h->uref_stride, h->rGuess);
            HIP_TRY(h, hipGetLastError());
            guess = h->rGuess;
        }
        if (h->sd.ready) {
            HIP_TRY(h, launch_sdual(h, 0, guess, o.polish_max_iter));
            // safety net (input box only, S = 0): what the dual method left without a certificate goes to the primal Riccati active set
            if (!h->sd.has_box && !h->sd.has_eq && !h->sd.useS && h->rKst) HIP_TRY(h, launch_riccati(h, 1, h->dU, 0));
        } else
            HIP_TRY(h, launch_riccati(h, 0, guess, o.polish_max_iter));
        h->r_has_step = true;
        return io_step_done();
    }
    const bool timing = (h->flags & ALMPC_FLAG_TIMING) != 0 && (h->step_count++ % (size_t)h->timing_stride) == 0;
    hipStream_t st = h->stream;
    hipEvent_t* ev = nullptr;
    if (timing) {
        if (h->ev.size() < 4 * (h->ev_used + 1)) {
            const size_t old = h->ev.size();
            h->ev.resize(old + 4 * 64, nullptr);
            for (size_t i = old; i < h->ev.size(); ++i) HIP_TRY(h, hipEventCreate(&h->ev[i]));
        }
        ev = &h->ev[4 * h->ev_used];
        if (h->ev_two.size() < h->ev_used + 1) h->ev_two.resize(h->ev_used + 64, 0);
    }
    // event 0 (start of the step) is recorded in front of the step's first launch -- unless the step turns out to be ONE kernel: then
    // only the pair around that kernel is recorded (an event record costs ~3.5 us of stream time: four of them are 5 % of such a step)
    bool ev0_done = false;
    auto ev0 = [&]() -> hipError_t {
        if (!timing || ev0_done) return hipSuccess;
        ev0_done = true;
        return hipEventRecord(ev[0], st);
    };

    AdmmInstParams ip;
    bool guess_ws = false;       // this step's guess came with the inverse of its working set (k_guess_iterate_ws)
    bool inst_pending = false;   // per-instance ADMM not launched yet: small problems go out fused with their finish (k_step_inst_wave)
    if (h->batched) {  // per-instance models: one workgroup per instance, KKT inverse in LDS
        ip.nz = h->nz; ip.n = h->n; ip.m = h->m; ip.batch = h->batch; ip.nzs = h->nzs;
        ip.Minv = h->bMinv; ip.Hs = h->bHs; ip.Fs = h->bFs; ip.Vs = h->bVs; ip.dvec = h->bD; ip.rhovec = h->bRho;
        ip.fS = h->dFS; ip.v0S = h->dV0S; ip.umin = h->dUmin; ip.umax = h->dUmax;
        ip.uref = h->dUref; ip.uref_stride = h->uref_stride; ip.xref = h->dXref; ip.xref_stride = h->xref_stride; ip.x0 = h->dX0;
        ip.xs = h->dXs; ip.zs = h->dZs; ip.ys = h->dYs; ip.v0 = h->dV0; ip.status = h->dStatus; ip.iters = h->dIters;
        ip.piters = h->dPiters; ip.perm = h->dPerm;
        ip.sigma = o.sigma; ip.alpha = o.alpha; ip.eps_abs = o.eps_abs; ip.eps_rel = o.eps_rel;
        ip.max_iter = o.max_iter; ip.check_every = o.check_every; ip.warm = o.warm_start ? 1 : 0;
        ip.mat_stride = (long)h->nz * h->nzs; ip.fv_stride = (long)h->n * h->nzs; ip.vec_stride = h->nzs; ip.fs_stride = h->nz;
        if (h->skip_admm) {
            if (!o.polish) return fail(h, ALMPC_ERR_INVALID, "calculate: the SQP loop needs opts.polish = 1");
            HIP_TRY(h, ev0());
            if (h->skip_admm == 2) hipLaunchKernelGGL(k_guess_shift, dim3((h->batch + 3) / 4), dim3(256), 0, st, ip, (const double*)h->dU, h->N);
            else if (h->nzs <= 128 && h->nzs > 64 && h->batch <= 2 * h->num_cus && h->dSglobal && h->bG && !getenv("ALMPC_NO_GUESS_WS")) {
                // the guess of an SQP iteration AND the inverse of its working set (33..64 of the inputs on a bound), four waves per
                // instance: k_guess_iterate_ws; installed by k_polish_sgl<1> below
                if (!h->dStartRows) HIP_TRY(h, dalloc(&h->dStartRows, (size_t)h->batch * 65));
                GuessWsParams gw;
                gw.G = h->bG; gw.G_stride = (long)h->nz * h->nzs; gw.sinv = h->dSglobal; gw.rows = h->dStartRows;
                hipLaunchKernelGGL(k_guess_iterate_ws, dim3((unsigned)h->batch), dim3(256), 0, st, ip, gw);
                guess_ws = true;
            } else hipLaunchKernelGGL(k_guess_iterate, dim3((h->batch + 3) / 4), dim3(256), 0, st, ip);
            HIP_TRY(h, hipGetLastError());
        } else {
            inst_pending = true;
        }
    }
    auto flush_admm_inst = [&]() -> int {   // the two-launch path of per-instance models: k_admm_inst on its own
        if (!inst_pending) return ALMPC_OK;
        inst_pending = false;
        HIP_TRY(h, ev0());
        const bool packed = h->minv_packed;
        const size_t l = admm_inst_lds_doubles(h->nz, h->nzs, h->m, packed) * sizeof(double);
        if (l > 64 * 1024)
            HIP_TRY(h, ensure_dyn_lds(packed ? reinterpret_cast<const void*>(k_admm_inst<true>) : reinterpret_cast<const void*>(k_admm_inst<false>), (size_t)(l)));
        // persistent grid: as many workgroups as fit the CUs at once (LDS bound; 512 threads each, at most 4 per CU)
        const int per_cu = 2;  // register bound: 2 x 256 threads at up to 256 VGPRs each fill the CU's register file
        int wgs = h->num_cus * per_cu;
        if (wgs > h->batch) wgs = h->batch;
        ip.minv_stride = packed ? packed_tri_doubles(h->nz) : 0;
        if (packed) hipLaunchKernelGGL((k_admm_inst<true>), dim3(wgs), dim3(ADMM_INST_THREADS), l, st, ip);
        else hipLaunchKernelGGL((k_admm_inst<false>), dim3(wgs), dim3(ADMM_INST_THREADS), l, st, ip);
        HIP_TRY(h, hipGetLastError());
        return ALMPC_OK;
    };
    AdmmParams ap;
    bool admm_pending = false;  // shared-model ADMM not launched yet: it may go out fused with the polish (k_step_fused)
    const int admm_grid = (h->batch + TILE - 1) / TILE;
    const size_t admm_lds = ((size_t)2 * h->nzs * TILE + (size_t)h->nrb * 8 * TILE + (size_t)4 * h->ksf * TILE) * sizeof(double);
    if (!h->batched) {
    ap.nz = h->nz; ap.n = h->n; ap.m = h->m; ap.batch = h->batch; ap.nzs = h->nzs;
    ap.MinvFrag = h->dMinvFrag; ap.VFrag = h->dVFrag; ap.v0S = h->dV0S; ap.v0S_stride = h->fS_stride; ap.HFrag = h->dHFrag; ap.FFrag = h->dFFrag; ap.ksf = h->ksf;
    ap.dvec = h->dD; ap.rhovec = h->dRho; ap.umin = h->dUmin; ap.umax = h->dUmax;
    ap.uref = h->dUref; ap.uref_stride = h->uref_stride; ap.xref = h->dXref; ap.xref_stride = h->xref_stride;
    ap.fS = h->dFS; ap.fS_stride = h->fS_stride; ap.x0 = h->dX0;
    ap.xs = h->dXs; ap.zs = h->dZs; ap.ys = h->dYs; ap.v0 = h->dV0; ap.status = h->dStatus; ap.iters = h->dIters;
    ap.piters = h->dPiters;
    ap.perm = h->dPerm;
    ap.rho = o.rho; ap.sigma = o.sigma; ap.alpha = o.alpha; ap.eps_abs = o.eps_abs; ap.eps_rel = o.eps_rel;
    ap.max_iter = o.max_iter; ap.check_every = o.check_every; ap.warm = o.warm_start ? 1 : 0;
    ap.keep_state = keep_state ? 1 : 0; ap.yflags = h->dYflags;
    admm_pending = true;
    }
    auto flush_admm = [&]() -> int {  // the two-kernel path: ADMM on its own
        { const int rci_ = flush_admm_inst(); if (rci_ != ALMPC_OK) return rci_; }
        HIP_TRY(h, ev0());
        if (admm_pending) {
            admm_pending = false;
            HIP_TRY(h, launch_admm(h->nrb, h->ks, ap, admm_grid, admm_lds, st));
        }
        if (timing) HIP_TRY(h, hipEventRecord(ev[1], st));
        return ALMPC_OK;
    };

    RolloutParams rp;
    rp.n = h->n; rp.m = h->m; rp.N = h->N; rp.batch = h->batch; rp.nzs = h->nzs; rp.A = h->batched ? h->bA : h->dA; rp.B = h->batched ? h->bB : h->dB;
    rp.dvec = h->dD; rp.w = h->dZs; rp.x0 = h->dX0; rp.xref = h->dXref; rp.xref_stride = h->xref_stride;
    rp.uref = h->dUref; rp.uref_stride = h->uref_stride; rp.umin = h->dUmin; rp.umax = h->dUmax; rp.x = h->dX; rp.ex = h->dEx; rp.u = h->dU; rp.eu = h->dEu;
    bool fused = false;
    if (h->mc > 0 && !o.polish)
        return fail(h, ALMPC_ERR_UNSUPPORTED, "calculate: problems with state rows (state box / terminal equality) need opts.polish = 1");
    int roll_g = 1;
    while (2 * roll_g * h->n <= 64) roll_g *= 2;
    const int roll_C = h->n + h->m;
    int roll_cpl = (roll_C + roll_g - 1) / roll_g;
    const bool roll_fits = (size_t)(h->N + 1) * roll_C <= 32 * 32 && roll_cpl <= 8;
    roll_cpl = roll_cpl <= 1 ? 1 : (roll_cpl <= 2 ? 2 : (roll_cpl <= 4 ? 4 : 8));
    if (h->mc > 0) {
        if (!roll_fits) return fail(h, ALMPC_ERR_UNSUPPORTED, "calculate: state rows need the fused rollout (n + m <= 8 * lanes-per-row)");
        { const int rc_ = flush_admm(); if (rc_ != ALMPC_OK) return rc_; }
        PolishGenParams gp;
        gp.nz = h->nz; gp.mc = h->mc; gp.R = h->R; gp.Rs = h->Rs; gp.m = h->m; gp.n = h->n; gp.N = h->N; gp.batch = h->batch; gp.nzs = h->nzs;
        if (h->batched && !h->ghat_inst) return fail(h, ALMPC_ERR_NOT_DESIGNED, "calculate: state rows without their per-instance matrices");
        if (h->ltv) {
            const bool sq = h->sqp.ready;
            gp.ltv = 1;
            gp.ltvA = sq ? h->sqp.A : h->lA; gp.ltvB = sq ? h->sqp.B : h->lB; gp.ltvC = sq ? h->sqp.c : h->lC;
            if (!gp.ltvA || !gp.ltvB) return fail(h, ALMPC_ERR_NOT_DESIGNED, "calculate: state rows of a time-varying design without its stage models");
            if (h->terminal_eq) { gp.eq_off = (sq ? h->sqp.ebar : h->lE) + (size_t)(h->N - 1) * h->n; gp.eq_stride = (long)h->N * h->n; }
        }
        if (h->batched) {
            gp.Ghat_stride = (long)h->R * h->Rs; gp.gnorm_stride = h->Rs; gp.d_stride = h->nzs;
            gp.A_stride = (long)h->n * h->n; gp.B_stride = (long)h->n * h->m;
        }
        if (!h->batched && h->roll_s > 0) { gp.rollM = h->dRollM; gp.roll_s = h->roll_s; gp.roll_nb = h->roll_nb; }
        if (h->eq_proj && !h->batched) { gp.eq_proj = 1; gp.eq0 = h->R - h->n; gp.ne = h->n; gp.GhatE = h->dGhatE; gp.WinvE = h->dWinvE; }
        gp.Ghat = h->dGhat; gp.gnorm = h->dGnorm; gp.row_traj = h->dRowTraj; gp.row_eq = h->dRowEq; gp.row_xidx = h->dRowXidx;
        gp.row_state = h->dRowState; gp.xmin = h->dXmin; gp.xmax = h->dXmax; gp.has_box = h->has_box;
        gp.dvec = h->batched ? h->bD : h->dD; gp.umin = h->dUmin; gp.umax = h->dUmax; gp.uref = h->dUref; gp.uref_stride = h->uref_stride;
        gp.zs = h->dZs; gp.ys = h->dYs; gp.v0 = h->dV0; gp.status = h->dStatus; gp.piters = h->dPiters;
        gp.max_iter = o.polish_max_iter > 0 ? o.polish_max_iter : 20 * h->R + 50;
        gp.roll_g = roll_g; gp.roll_cpl = roll_cpl; gp.roll = rp;
        if (!h->dOverflow) HIP_TRY(h, dalloc(&h->dOverflow, (size_t)h->batch * 33 + 2));   // list, then [batch][32] working sets
        HIP_TRY(h, hipMemsetAsync(h->dOverflow, 0, 2 * sizeof(int32_t), st));
        if (!h->dOvfSinv) HIP_TRY(h, dalloc(&h->dOvfSinv, (size_t)h->batch * (32 * 32 + 32)));
        gp.ovf = h->dOverflow; gp.ovf_ws = h->dOverflow + 2 + h->batch; gp.ovf_sinv = h->dOvfSinv;
        if (lazy_redo) { HIP_TRY(h, ensure_unsolved_word()); gp.unsolved = h->dUnsolved; gp.redo_gate = h->dRedoGate; gp.step_serial = h->step_serial; }
        if (h->fallback && !h->ltv && h->sd.ready && !h->sd.sqp && !getenv("ALMPC_NO_REDO_START")) {   // a stage-wise redo may follow: it starts from what this finish gives up with
            if (!h->sd.start_ws) HIP_TRY(h, dalloc(&h->sd.start_ws, (size_t)h->batch * 64));
            gp.redo_ws = h->sd.start_ws; gp.redo_sp = h->sd.NT + h->sd.MC; gp.redo_nt = h->sd.NT;
            h->sd.start_ws_fresh = true;
        }
        if (h->s0_basis_ok && !h->batched && !h->ltv && h->fS_stride == 0) gp.s0_basis = h->dS0Basis;   // (shared model, shared references)
        const dim3 grid((h->batch + PGEN_WAVES - 1) / PGEN_WAVES), block(64 * PGEN_WAVES);
        // first launch: working sets up to 32 rows, every instance; second launch: the instances the first one flagged, up to 64 rows
        const size_t l32 = (size_t)PGEN_WAVES * pgen_lds_per_wave(32) * sizeof(double);
        const size_t l64 = (size_t)pgen_coop_lds_doubles() * sizeof(double);   // (wave 0's buffers + job word + three partial sums)
#define PGEN_LAUNCH(NP_)                                                                                                         \
    do {                                                                                                                         \
        hipLaunchKernelGGL((k_polish_gen<NP_>), grid, block, l32, st, gp);                                                       \
        HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(k_polish_gen64<NP_>), l64));                                     \
        hipLaunchKernelGGL((k_polish_gen64<NP_>), dim3(3 * h->num_cus < h->batch ? 3 * h->num_cus : h->batch), dim3(64 * (PGEN_HELPERS + 1)), l64, st, gp); \
    } while (0)
        switch (h->np_pairs) {
            case 1: PGEN_LAUNCH(1); break;
            case 2: PGEN_LAUNCH(2); break;
            case 3: PGEN_LAUNCH(3); break;
            default: PGEN_LAUNCH(4); break;
        }
#undef PGEN_LAUNCH
        HIP_TRY(h, hipGetLastError());
        fused = true;
    } else if (o.polish) {
        PolishParams pp;
        pp.nz = h->nz; pp.m = h->m; pp.batch = h->batch; pp.nzs = h->nzs;
        pp.G = h->batched ? h->bG : h->dG; pp.dvec = h->batched ? h->bD : h->dD;
        pp.G_stride = h->batched ? (long)h->nz * h->nzs : 0; pp.d_stride = h->batched ? (long)h->nzs : 0;
        pp.A_stride = h->batched ? (long)h->n * h->n : 0; pp.B_stride = h->batched ? (long)h->n * h->m : 0;
        pp.wave_const_off = -1; pp.sg_off = 0; pp.g_off = 0; pp.sg_shared_off = -1;
        pp.umin = h->dUmin; pp.umax = h->dUmax; pp.uref = h->dUref; pp.uref_stride = h->uref_stride;
        pp.zs = h->dZs; pp.ys = h->dYs; pp.v0 = h->dV0; pp.w = h->dW; pp.status = h->dStatus; pp.piters = h->dPiters;
        pp.yflags = keep_state ? nullptr : h->dYflags; pp.yflag_words = h->nrb;
        pp.sglobal = h->dSglobal; pp.perm = h->dPerm; pp.ntiles = (h->batch + 15) / 16;
        pp.max_iter = o.polish_max_iter > 0 ? o.polish_max_iter : 2 * h->nz + 50;
        pp.dflag = h->flag_in_finish ? h->bFlag : nullptr;
        if (lazy_redo) {
            HIP_TRY(h, ensure_unsolved_word());
            pp.unsolved = h->dUnsolved; pp.redo_gate = h->dRedoGate; pp.step_serial = h->step_serial;
        }
        // rollout fused into the tail of the polish when its trajectory buffer fits the wave's LDS slot
        fused = roll_fits;
        const bool blocked = !h->batched && h->roll_s > 0;   // shared model: blocked rollout, no trajectory buffer
        if (blocked) fused = true;
        pp.fuse_rollout = fused ? (h->ltv ? 2 : (blocked ? 3 : 1)) : 0; pp.roll_g = roll_g; pp.roll_cpl = roll_cpl; pp.roll = rp;
        pp.rollM = h->dRollM; pp.roll_s = h->roll_s; pp.roll_nb = h->roll_nb;
        int per_wave = POLISH_LDS_MIN_PER_WAVE;
        if (fused && !blocked && (h->N + 1) * (h->n + h->m) > per_wave) per_wave = (h->N + 1) * (h->n + h->m);
        per_wave = (per_wave + 1) & ~1;
        if (h->batched) {  // the wave's private copy of d_i | [A_i B_i] sits behind its buffers
            if (!fused) return fail(h, ALMPC_ERR_UNSUPPORTED, "calculate: per-instance models need the fused rollout (n + m <= 8 * lanes-per-row)");
            pp.wave_const_off = per_wave;
            per_wave += (h->nzs + h->n * (h->n + h->m) + 1) & ~1;
        }
        pp.lds_per_wave = per_wave;
        // G in LDS when it fits beside the buffers of 8 waves (gfx950: 160 KB per workgroup)
        const PolishShared SL = polish_shared_layout(h->n, h->m, h->N, h->nz, h->nzs, pp.fuse_rollout);
        const size_t g_lds = (size_t)h->nz * ((h->nz + 1) & ~1);  // doubles: rows of G packed to an even stride
        size_t l_glds = (g_lds + SL.total + (size_t)POLISH_WAVES_GLDS * per_wave + 2) * sizeof(double);
        // workgroup-shared second-tier slot (working sets beyond 32 rows) behind the queue words, if the 160 KB allow it
        pp.sg_shared_off = -1;
        {
            const size_t slot = (size_t)POLISH_SG_SHARED_CAP * 64 * sizeof(double);
            if (l_glds + slot <= 160 * 1024 && !h->batched && !getenv("ALMPC_POLISH_SG_GLOBAL")) {
                pp.sg_shared_off = (int)(SL.total + (size_t)POLISH_WAVES_GLDS * per_wave + 2);
                l_glds += slot;
            }
        }
        // one kernel for the whole step when the tile is 8 waves and [G | union(ADMM buffers, polish buffers)] fits LDS
        size_t l_step = l_glds - g_lds * sizeof(double);
        if (admm_lds > l_step) l_step = admm_lds;
        l_step += g_lds * sizeof(double);
        const bool step_fused = admm_pending && h->fuse_step && POLISH_WAVES_GLDS == 8 && !h->polish_no_glds && h->nrb == 8 && (h->ks == 30 || h->ks == 32) &&
                                fused && l_step <= 160 * 1024;
        if (step_fused) {
            admm_pending = false;
            if (timing) HIP_TRY(h, hipEventRecord(ev[1], st));  // no boundary between the phases to time: admm_ms reads 0
            if (h->ks == 30) {
                HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(k_step_fused<8, 30>), (size_t)(l_step)));
                hipLaunchKernelGGL((k_step_fused<8, 30>), dim3(pp.ntiles), dim3(512), l_step, st, ap, pp);
            } else {
                HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(k_step_fused<8, 32>), (size_t)(l_step)));
                hipLaunchKernelGGL((k_step_fused<8, 32>), dim3(pp.ntiles), dim3(512), l_step, st, ap, pp);
            }
                } else if (admm_pending && !h->batched && h->nzs <= 64 && fused && h->dPlain && !h->ltv &&
                           !(getenv("ALMPC_NO_SHARED_WAVE") && getenv("ALMPC_NO_SHARED_WAVE")[0] == '1') &&
                           (long)h->batch <= (getenv("ALMPC_SHARED_WAVE_MAX_BATCH") ? atol(getenv("ALMPC_SHARED_WAVE_MAX_BATCH")) : (long)2 * h->num_cus) &&
                           ((size_t)SL.total + (size_t)per_wave) * sizeof(double) <= 64 * 1024) {
            // small SHARED problems in small batches (configs[0], the reference's own test sizes: one instance, N 5 - 15, m 2 -> nz
            // 10 - 30; round 5): the two-launch path gives them a 16-instance MFMA tile they cannot fill and two launch ramps.  ONE
            // kernel, one wave per instance -- k_step_inst_wave with operand strides of zero: every wave reads the same dense
            // Minv / F' / V (a few KB, L1 / L2 resident), runs the ADMM iterations from registers and then the single-wave finish
            // (G through L1: 13 KB at nz 40).  Measured (tools/time_small_shared.py, QTP fixture, N 5 / N 20, us per step): batch 1
            // 16.9 / 21.8 against 29.7 / 33.1, batch 64 20 / 45 against 43 / 56, batch 512 30 / 58 against 44 / 58; from 2048
            // instances on a wave per 10 - 40-row problem wastes the machine (63 / 113 against 45 / 70 us; 65,536: 1.8 / 2.3 ms
            // against 0.32 / 0.68 ms): up to two instances per CU take this path, larger batches the tile path.
            admm_pending = false;
            HIP_TRY(h, ev0());
            const size_t mm = (size_t)h->nz * h->nzs, fv = (size_t)h->n * h->nzs;
            ip.nz = h->nz; ip.n = h->n; ip.m = h->m; ip.batch = h->batch; ip.nzs = h->nzs;
            ip.Minv = h->dPlain; ip.Hs = h->dPlain + mm; ip.Fs = h->dPlain + 2 * mm; ip.Vs = h->dPlain + 2 * mm + fv;
            ip.dvec = h->dD; ip.rhovec = h->dRho;
            ip.mat_stride = 0; ip.fv_stride = 0; ip.vec_stride = 0; ip.fs_stride = h->fS_stride;
            ip.fS = h->dFS; ip.v0S = h->dV0S; ip.umin = h->dUmin; ip.umax = h->dUmax;
            ip.uref = h->dUref; ip.uref_stride = h->uref_stride; ip.xref = h->dXref; ip.xref_stride = h->xref_stride; ip.x0 = h->dX0;
            ip.xs = h->dXs; ip.zs = h->dZs; ip.ys = h->dYs; ip.v0 = h->dV0; ip.status = h->dStatus; ip.iters = h->dIters;
            ip.piters = h->dPiters; ip.perm = h->dPerm;
            ip.sigma = o.sigma; ip.alpha = o.alpha; ip.eps_abs = o.eps_abs; ip.eps_rel = o.eps_rel;
            ip.max_iter = o.max_iter; ip.check_every = o.check_every; ip.warm = o.warm_start ? 1 : 0;
            pp.yflags = nullptr; pp.yflag_words = 0;   // (this ADMM phase hands over y itself)
            const size_t l_sgl = ((size_t)SL.total + (size_t)per_wave) * sizeof(double);
            pp.sg_off = -1; pp.g_off = 0;
            pp.lds_per_wave = per_wave;
            pp.direct = 1;
            if (timing) HIP_TRY(h, hipEventRecord(ev[1], st));
#define STEP_SHARED(NZC_)                                                                                            \
    do {                                                                                                             \
        HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(k_step_inst_wave<NZC_>), l_sgl));                    \
        hipLaunchKernelGGL((k_step_inst_wave<NZC_>), dim3((unsigned)h->batch), dim3(64), l_sgl, st, ip, pp);         \
    } while (0)
            if (h->nzs <= 16) STEP_SHARED(16);
            else if (h->nzs <= 32) STEP_SHARED(32);
            else if (h->nzs <= 48) STEP_SHARED(48);
            else STEP_SHARED(64);
#undef STEP_SHARED
                } else if (inst_pending && h->nzs <= 64 && fused && !(getenv("ALMPC_NO_INST_WAVE") && getenv("ALMPC_NO_INST_WAVE")[0] == '1') &&
                           ((size_t)SL.total + (size_t)per_wave + (size_t)h->nz * h->nzs) * sizeof(double) <= 64 * 1024) {
            // small per-instance problems (BASELINE configs[3]): ONE wave per instance for the whole step -- ADMM with the KKT
            // inverse in registers, then the single-wave finish with G_i in the wave's LDS (the second-tier Sinv, rarely needed at
            // these sizes, stays in the global scratch: with its 32 KB per wave only three waves would fit a CU)
            inst_pending = false;
            const size_t sgl_wave = (size_t)per_wave + (size_t)h->nz * h->nzs;
            const size_t l_sgl = ((size_t)SL.total + sgl_wave) * sizeof(double);
            pp.sg_off = -1;
            pp.g_off = per_wave;
            pp.lds_per_wave = (int)sgl_wave;
            pp.direct = 1;
            if (timing) HIP_TRY(h, hipEventRecord(ev[1], st));  // no boundary between the phases to time: admm_ms reads 0
#define STEP_INST(NZC_)                                                                                              \
    do {                                                                                                             \
        HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(k_step_inst_wave<NZC_>), l_sgl));                    \
        hipLaunchKernelGGL((k_step_inst_wave<NZC_>), dim3((unsigned)h->batch), dim3(64), l_sgl, st, ip, pp);         \
    } while (0)
            if (h->nzs <= 16) STEP_INST(16);
            else if (h->nzs <= 32) STEP_INST(32);
            else if (h->nzs <= 48) STEP_INST(48);
            else STEP_INST(64);
#undef STEP_INST
                } else {
        { const int rc_ = flush_admm(); if (rc_ != ALMPC_OK) return rc_; }
        if (l_glds <= 160 * 1024 && !h->polish_no_glds && !h->batched) {
            HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(k_polish<true>), (size_t)(l_glds)));
            int wgs = pp.ntiles;  // one ADMM tile (16 instances) per workgroup and round
            if (wgs > h->num_cus) wgs = h->num_cus;
            hipLaunchKernelGGL((k_polish<true>), dim3(wgs), dim3(64 * POLISH_WAVES_GLDS), l_glds, st, pp);
        } else {
            const size_t l = ((size_t)SL.total + (size_t)POLISH_WAVES * per_wave) * sizeof(double);
            // small batches of per-instance models: single-wave workgroups with G_i and the second-tier Sinv in LDS
            const size_t sgl_wave = (size_t)per_wave + POLISH_GLB_PER_INST + (size_t)h->nz * h->nzs;
            const size_t l_sgl = ((size_t)SL.total + sgl_wave) * sizeof(double);
            if (h->batched && h->batch <= 2 * h->num_cus && l_sgl <= 160 * 1024 && !getenv("ALMPC_POLISH_SG_GLOBAL")) {
                pp.sg_off = per_wave;
                pp.g_off = per_wave + POLISH_GLB_PER_INST;
                pp.lds_per_wave = (int)sgl_wave;
                if (guess_ws) {
                    pp.start_rows = h->dStartRows;
                    HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(k_polish_sgl<1>), (size_t)(l_sgl)));
                    hipLaunchKernelGGL((k_polish_sgl<1>), dim3(pp.ntiles * 16), dim3(64), l_sgl, st, pp);
                } else {
                HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(k_polish_sgl<0>), (size_t)(l_sgl)));
                hipLaunchKernelGGL((k_polish_sgl<0>), dim3(pp.ntiles * 16), dim3(64), l_sgl, st, pp);
                }
            } else {
            HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(k_polish<false>), l));
            hipLaunchKernelGGL((k_polish<false>), dim3((pp.ntiles * 16 + POLISH_WAVES - 1) / POLISH_WAVES), dim3(64 * POLISH_WAVES), l, st, pp);
            }
        }
        }
        HIP_TRY(h, hipGetLastError());
        rp.w = h->dW;
    } else {
        const int rc_ = flush_admm();
        if (rc_ != ALMPC_OK) return rc_;
    }
    if (timing) HIP_TRY(h, hipEventRecord(ev[2], st));

    if (!fused) {
        const size_t per_wave = (size_t)h->n * (h->N + 1) + h->nz, shared = (size_t)h->n * h->n + (size_t)h->n * h->m;
        if (h->batched) { rp.A_stride = (long)h->n * h->n; rp.B_stride = (long)h->n * h->m; rp.d_stride = h->nzs; rp.dvec = h->bD; }
        if (!h->batched && (shared + 4 * per_wave) * sizeof(double) <= 60 * 1024) {
            const size_t l = (shared + 4 * per_wave) * sizeof(double);
            hipLaunchKernelGGL((k_rollout<4>), dim3((h->batch + 3) / 4), dim3(256), l, st, rp);
        } else {  // long horizons with many states: one instance per workgroup, LDS beyond the 64 KiB default
            const size_t l = (shared + per_wave) * sizeof(double);
            HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(k_rollout<1>), (size_t)(l)));
            hipLaunchKernelGGL((k_rollout<1>), dim3(h->batch), dim3(64), l, st, rp);
        }
        HIP_TRY(h, hipGetLastError());
    }
    // structured fallback: instances the condensed path left without a certificate (status != 0: an active-set finish that ran into
    // its cap, a non-finite or indefinite condensed problem) are redone in the multiple-shooting form, from the step's own result
    if (lazy_redo) h->lazy_pending = true;   // (resolve_lazy_redo at the next host sync point)
    else if (h->fallback && !h->ltv && o.polish) {
        if (h->sd.ready && !getenv("ALMPC_DBG_NO_SDUAL_FB")) {
            if (h->sd.per_instance) HIP_TRY(h, launch_sgains(h, 1));   // (stage records of the unsolved instances only, from the models of this step)
            HIP_TRY(h, launch_sdual(h, 1, h->dU, 0, true));
        }
        if (h->mc == 0 && !h->useS && h->rKst && !getenv("ALMPC_DBG_NO_PRIMAL_NET")) HIP_TRY(h, launch_riccati(h, 1, h->dU, 0));
    }
    if (timing) {
        if (ev0_done) HIP_TRY(h, hipEventRecord(ev[3], st));
        h->ev_two[h->ev_used] = ev0_done ? 0 : 1;
        h->ev_used += 1;
    }
    return io_step_done();
}

int almpc_synchronize(almpc_handle* h) {
    if (!h) return ALMPC_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    // A step is tens of microseconds: the wait polls the stream for a while before falling back to the blocking wait, whose wake-up
    // (interrupt + scheduler) was seen to cost up to ~15 ms on this pool -- 200 steps' worth (stream_wait_polling).
    return wait_and_settle(h);
}

int almpc_calculate(almpc_handle* h, const almpc_opts* opts) {
    int rc = almpc_calculate_async(h, opts);
    if (rc != ALMPC_OK) return rc;
    return almpc_synchronize(h);
}

int almpc_get_results(almpc_handle* h, double* x, double* e_x, double* u, double* e_u, int32_t* status,
                      int32_t* iters, int32_t* polish_iters) {
    if (!h) return ALMPC_ERR_INVALID;
    if (!h->designed) return fail(h, ALMPC_ERR_NOT_DESIGNED, "get_results before design");
    HIP_TRY(h, hipSetDevice(h->device));
    { const int rc_ = wait_and_settle(h, true); if (rc_ != ALMPC_OK) return rc_; }
    const size_t b = (size_t)h->batch, xs = (size_t)h->n * (h->N + 1), us = (size_t)h->nz;
    if (x) HIP_TRY(h, hipMemcpy(x, h->dX, b * xs * sizeof(double), hipMemcpyDeviceToHost));
    if (e_x) HIP_TRY(h, hipMemcpy(e_x, h->dEx, b * xs * sizeof(double), hipMemcpyDeviceToHost));
    if (u) HIP_TRY(h, hipMemcpy(u, h->dU, b * us * sizeof(double), hipMemcpyDeviceToHost));
    if (e_u) HIP_TRY(h, hipMemcpy(e_u, h->dEu, b * us * sizeof(double), hipMemcpyDeviceToHost));
    if (status) HIP_TRY(h, hipMemcpy(status, h->dStatus, b * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (iters) HIP_TRY(h, hipMemcpy(iters, h->dIters, b * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (polish_iters) HIP_TRY(h, hipMemcpy(polish_iters, h->dPiters, b * sizeof(int32_t), hipMemcpyDeviceToHost));
    return ALMPC_OK;
}

int almpc_get_design(almpc_handle* h, double* H, double* F, double* P, double* d) {
    if (!h) return ALMPC_ERR_INVALID;
    if (!h->designed) return fail(h, ALMPC_ERR_NOT_DESIGNED, "get_design before design");
    if (H) std::memcpy(H, h->H.data(), h->H.size() * sizeof(double));
    if (F) std::memcpy(F, h->F.data(), h->F.size() * sizeof(double));
    if (P) std::memcpy(P, h->P.data(), h->P.size() * sizeof(double));
    if (d && h->d.size() >= (size_t)h->nz) std::memcpy(d, h->d.data(), (size_t)h->nz * sizeof(double));   // (structured handles: P only)
    return ALMPC_OK;
}

int almpc_device_results(almpc_handle* h, const double** d_x, const double** d_e_x, const double** d_u,
                         const double** d_e_u) {
    if (!h) return ALMPC_ERR_INVALID;
    if (d_x) *d_x = h->dX;
    if (d_e_x) *d_e_x = h->dEx;
    if (d_u) *d_u = h->dU;
    if (d_e_u) *d_e_u = h->dEu;
    return ALMPC_OK;
}

int almpc_timing_set_stride(almpc_handle* h, int every) {
    if (!h || every < 1) return ALMPC_ERR_INVALID;
    h->timing_stride = every;
    h->step_count = 0;
    return ALMPC_OK;
}

int almpc_timing_reset(almpc_handle* h, int reserve_steps) {
    if (!h) return ALMPC_ERR_INVALID;
    if (!(h->flags & ALMPC_FLAG_TIMING)) return fail(h, ALMPC_ERR_INVALID, "handle was created without ALMPC_FLAG_TIMING");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->ev_used = 0;
    h->step_count = 0;
    const size_t need = 4 * (size_t)(reserve_steps > 0 ? reserve_steps : 0);
    if (h->ev.size() < need) {
        const size_t old = h->ev.size();
        h->ev.resize(need, nullptr);
        for (size_t i = old; i < need; ++i) HIP_TRY(h, hipEventCreate(&h->ev[i]));
    }
    return ALMPC_OK;
}

int almpc_timing_summary(almpc_handle* h, int* steps, double* ms_admm, double* ms_polish, double* ms_rollout,
                         double* ms_total) {
    if (!h) return ALMPC_ERR_INVALID;
    if (!(h->flags & ALMPC_FLAG_TIMING)) return fail(h, ALMPC_ERR_INVALID, "handle was created without ALMPC_FLAG_TIMING");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    double a = 0, p = 0, r = 0, t = 0;
    for (size_t sidx = 0; sidx < h->ev_used; ++sidx) {
        hipEvent_t* ev = &h->ev[4 * sidx];
        float f = 0;
        if (h->ev_two[sidx]) {   // one-kernel step: the pair around the kernel is all there is
            HIP_TRY(h, hipEventElapsedTime(&f, ev[1], ev[2])); p += f; t += f;
            continue;
        }
        HIP_TRY(h, hipEventElapsedTime(&f, ev[0], ev[1])); a += f;
        HIP_TRY(h, hipEventElapsedTime(&f, ev[1], ev[2])); p += f;
        HIP_TRY(h, hipEventElapsedTime(&f, ev[2], ev[3])); r += f;
        HIP_TRY(h, hipEventElapsedTime(&f, ev[0], ev[3])); t += f;
    }
    if (steps) *steps = (int)h->ev_used;
    if (ms_admm) *ms_admm = a;
    if (ms_polish) *ms_polish = p;
    if (ms_rollout) *ms_rollout = r;
    if (ms_total) *ms_total = t;
    return ALMPC_OK;
}

int almpc_timing_samples(almpc_handle* h, int cap, int* count, float* ms_admm, float* ms_polish, float* ms_rollout, float* ms_total) {
    if (!h || cap < 0) return ALMPC_ERR_INVALID;
    if (!(h->flags & ALMPC_FLAG_TIMING)) return fail(h, ALMPC_ERR_INVALID, "handle was created without ALMPC_FLAG_TIMING");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t nrec = std::min((size_t)cap, h->ev_used);
    for (size_t sidx = 0; sidx < nrec; ++sidx) {
        hipEvent_t* ev = &h->ev[4 * sidx];
        float f = 0;
        if (h->ev_two[sidx]) {
            HIP_TRY(h, hipEventElapsedTime(&f, ev[1], ev[2]));
            if (ms_admm) ms_admm[sidx] = 0; if (ms_polish) ms_polish[sidx] = f; if (ms_rollout) ms_rollout[sidx] = 0; if (ms_total) ms_total[sidx] = f;
            continue;
        }
        HIP_TRY(h, hipEventElapsedTime(&f, ev[0], ev[1])); if (ms_admm) ms_admm[sidx] = f;
        HIP_TRY(h, hipEventElapsedTime(&f, ev[1], ev[2])); if (ms_polish) ms_polish[sidx] = f;
        HIP_TRY(h, hipEventElapsedTime(&f, ev[2], ev[3])); if (ms_rollout) ms_rollout[sidx] = f;
        HIP_TRY(h, hipEventElapsedTime(&f, ev[0], ev[3])); if (ms_total) ms_total[sidx] = f;
    }
    if (count) *count = (int)h->ev_used;
    return ALMPC_OK;
}

int almpc_get_timing(almpc_handle* h, float* ms_admm, float* ms_polish, float* ms_rollout, float* ms_total) {
    if (!h) return ALMPC_ERR_INVALID;
    if (!(h->flags & ALMPC_FLAG_TIMING) || h->ev_used == 0) return fail(h, ALMPC_ERR_INVALID, "timing not enabled or no step timed yet");
    HIP_TRY(h, hipSetDevice(h->device));
    hipEvent_t* ev = &h->ev[4 * (h->ev_used - 1)];
    float a = 0, p = 0, r = 0, t = 0;
    if (h->ev_two[h->ev_used - 1]) {
        HIP_TRY(h, hipEventSynchronize(ev[2]));
        HIP_TRY(h, hipEventElapsedTime(&p, ev[1], ev[2]));
        t = p;
    } else {
    HIP_TRY(h, hipEventSynchronize(ev[3]));
    HIP_TRY(h, hipEventElapsedTime(&a, ev[0], ev[1]));
    HIP_TRY(h, hipEventElapsedTime(&p, ev[1], ev[2]));
    HIP_TRY(h, hipEventElapsedTime(&r, ev[2], ev[3]));
    HIP_TRY(h, hipEventElapsedTime(&t, ev[0], ev[3]));
    }
    if (ms_admm) *ms_admm = a;
    if (ms_polish) *ms_polish = p;
    if (ms_rollout) *ms_rollout = r;
    if (ms_total) *ms_total = t;
    return ALMPC_OK;
}

int almpc_dare(int n, int m, const double* A, const double* B, const double* Q, const double* R, double* P) {
    if (n < 1 || m < 1 || !A || !B || !Q || !R || !P) return ALMPC_ERR_INVALID;
    hm::mat Am(A, A + (size_t)n * n), Bm(B, B + (size_t)n * m), Qm(Q, Q + (size_t)n * n), Rm(R, R + (size_t)m * m), Pm;
    if (!hm::dare(Am, Bm, Qm, Rm, n, m, Pm)) return ALMPC_ERR_NUMERIC;
    std::memcpy(P, Pm.data(), (size_t)n * n * sizeof(double));
    return ALMPC_OK;
}

int almpc_fnn_linearize(int device_id, int n, int m, int H, int L, int activation, const double* W_in, const double* W_h,
                        const double* b_h, const double* W_out, int batch, const double* x, const double* u, double* A,
                        double* B, double* f) {
    if (n < 1 || m < 1 || H < 1 || L < 0 || batch < 1 || !W_in || !W_out || !x || !u || !A || !B || (L > 0 && (!W_h || !b_h)))
        return ALMPC_ERR_INVALID;
    if (activation < 0 || activation > 4) return ALMPC_ERR_UNSUPPORTED;
    const size_t nin = (size_t)n + m;
    const size_t lds = (2 * (size_t)H + 2 * (size_t)H * nin + nin) * sizeof(double);
    if (lds > 160 * 1024) return ALMPC_ERR_UNSUPPORTED;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device_id < 0 || device_id >= count) return ALMPC_ERR_NO_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) return ALMPC_ERR_HIP;
    std::vector<void*> bufs;
    auto up = [&](const double* src, size_t cnt) -> double* {
        double* d = nullptr;
        if (cnt == 0) cnt = 1;
        if (hipMalloc(reinterpret_cast<void**>(&d), cnt * sizeof(double)) != hipSuccess) return nullptr;
        bufs.push_back(d);
        if (src && hipMemcpy(d, src, cnt * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return d;
    };
    FnnParams p;
    p.n = n; p.m = m; p.H = H; p.L = L; p.act = activation; p.batch = batch;
    p.ppi = 1; p.xs_group = n; p.us_group = m;
    p.W_in = up(W_in, (size_t)H * nin); p.W_h = up(W_h, (size_t)L * H * H); p.b_h = up(b_h, (size_t)L * H);
    p.W_out = up(W_out, (size_t)n * H); p.x = up(x, (size_t)batch * n); p.u = up(u, (size_t)batch * m);
    p.A = up(nullptr, (size_t)batch * n * n); p.B = up(nullptr, (size_t)batch * n * m); p.f = f ? up(nullptr, (size_t)batch * n) : nullptr;
    int rc = ALMPC_OK;
    if (!p.W_in || !p.W_h || !p.b_h || !p.W_out || !p.x || !p.u || !p.A || !p.B || (f && !p.f)) rc = ALMPC_ERR_HIP;
    if (rc == ALMPC_OK) {
        hipDeviceProp_t prop;
        const int cus = hipGetDeviceProperties(&prop, device_id) == hipSuccess ? prop.multiProcessorCount : 256;
        if (launch_fnn_jacobian(p, cus, 0) != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = ALMPC_ERR_HIP;
    }
    if (rc == ALMPC_OK) {
        if (hipMemcpy(A, p.A, (size_t)batch * n * n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(B, p.B, (size_t)batch * n * m * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
            (f && hipMemcpy(f, p.f, (size_t)batch * n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess))
            rc = ALMPC_ERR_HIP;
    }
    for (void* b : bufs) (void)hipFree(b);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Multi-GPU: RCCL inside the library (see almpc_comm.hip.h and include/almpc.h)
int almpc_comm_unique_id(char* id128) {
    if (!id128) return ALMPC_ERR_INVALID;
    RcclApi& r = rccl_api();
    if (!r.ok) return ALMPC_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return ALMPC_ERR_HIP;
    std::memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return ALMPC_OK;
}

int almpc_comm_init(almpc_handle* h, const char* id128, int rank, int world) {
    if (!h || !id128) return ALMPC_ERR_INVALID;
    if (world < 1 || rank < 0 || rank >= world) return fail(h, ALMPC_ERR_INVALID, "comm_init: need 0 <= rank < world");
    RcclApi& r = rccl_api();
    if (!r.ok) return fail(h, ALMPC_ERR_UNSUPPORTED, "comm_init: " + r.err);
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->comm) { (void)r.CommDestroy(h->comm); h->comm = nullptr; }
    ncclUniqueId id;
    std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    const ncclResult_t rc = r.CommInitRank(&h->comm, world, id, rank);
    if (rc != ncclSuccess) { h->comm = nullptr; return fail(h, ALMPC_ERR_HIP, std::string("ncclCommInitRank: ") + r.GetErrorString(rc)); }
    h->comm_rank = rank; h->comm_world = world;
    for (double** pp_ : {&h->dU0, &h->dU0all})
        if (*pp_) { (void)hipFree(*pp_); *pp_ = nullptr; }
    if (!h->dComm4) HIP_TRY(h, dalloc(&h->dComm4, 4));
    HIP_TRY(h, dalloc(&h->dU0, (size_t)h->batch * h->m));
    HIP_TRY(h, dalloc(&h->dU0all, (size_t)world * h->batch * h->m));
    return ALMPC_OK;
}

int almpc_comm_summary(almpc_handle* h, int64_t* out4) {
    if (!h || !out4) return ALMPC_ERR_INVALID;
    if (!h->comm) return fail(h, ALMPC_ERR_NOT_DESIGNED, "comm_summary before comm_init");
    if (!h->designed) return fail(h, ALMPC_ERR_NOT_DESIGNED, "comm_summary before a step");
    RcclApi& r = rccl_api();
    HIP_TRY(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(k_comm_summary, dim3(1), dim3(1024), 0, h->stream, h->batch, h->dStatus, h->dIters, h->dPiters, h->dComm4);
    HIP_TRY(h, hipGetLastError());
    ncclResult_t rc = r.AllReduce(h->dComm4, h->dComm4, 2, ncclInt64, ncclSum, h->comm, h->stream);
    if (rc == ncclSuccess) rc = r.AllReduce(h->dComm4 + 2, h->dComm4 + 2, 2, ncclInt64, ncclMax, h->comm, h->stream);
    if (rc != ncclSuccess) return fail(h, ALMPC_ERR_HIP, std::string("ncclAllReduce: ") + r.GetErrorString(rc));
    long long host[4];
    HIP_TRY(h, hipMemcpyAsync(host, h->dComm4, sizeof(host), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 4; ++i) out4[i] = host[i];
    return ALMPC_OK;
}

int almpc_comm_allgather_first_input(almpc_handle* h, double* u0_all, const double** d_u0_all) {
    if (!h) return ALMPC_ERR_INVALID;
    if (!h->comm) return fail(h, ALMPC_ERR_NOT_DESIGNED, "comm_allgather_first_input before comm_init");
    if (!h->designed) return fail(h, ALMPC_ERR_NOT_DESIGNED, "comm_allgather_first_input before a step");
    RcclApi& r = rccl_api();
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t cnt = (size_t)h->batch * h->m;
    hipLaunchKernelGGL(k_pack_first_input, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, h->batch, h->m, h->N, h->dU, h->dU0);
    HIP_TRY(h, hipGetLastError());
    const ncclResult_t rc = r.AllGather(h->dU0, h->dU0all, cnt, ncclFloat64, h->comm, h->stream);
    if (rc != ncclSuccess) return fail(h, ALMPC_ERR_HIP, std::string("ncclAllGather: ") + r.GetErrorString(rc));
    if (u0_all) {
        HIP_TRY(h, hipMemcpyAsync(u0_all, h->dU0all, cnt * h->comm_world * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    if (d_u0_all) *d_u0_all = h->dU0all;
    return ALMPC_OK;
}

int almpc_advance_plant(almpc_handle* h) {
    if (!h) return ALMPC_ERR_INVALID;
    if (!h->designed) return fail(h, ALMPC_ERR_NOT_DESIGNED, "advance_plant before design");
    if (h->batched) return fail(h, ALMPC_ERR_UNSUPPORTED, "advance_plant: per-instance models have no shared plant (advance the states on the caller's side)");
    HIP_TRY(h, hipSetDevice(h->device));
    { const int rc_ = enqueue_gated_redo(h); if (rc_ != ALMPC_OK) return rc_; }   // the plant must not be driven by an undecided instance's iterate
    const int per_block = 256 / h->n;
    const double* x0_in = h->dX0;
    const int x0_slot = h->io.x0_slot;
    io_release_x0(h);   // (a pinned x0 slot is read once more here; the new states go to the handle's own device buffer)
    hipLaunchKernelGGL(k_advance_plant, dim3((h->batch + per_block - 1) / per_block), dim3(256), (size_t)per_block * h->n * sizeof(double),
                       h->stream, h->n, h->m, h->N, h->batch, h->dA, h->dB, h->dU, x0_in, h->dX0);
    HIP_TRY(h, hipGetLastError());
    if (x0_slot >= 0) {   // the slot is free for the host again only once THIS read has finished too
        HIP_TRY(h, hipEventRecord(h->io.ev_used[x0_slot], h->stream));
        h->io.used_pending[x0_slot] = true;
    }
    return ALMPC_OK;
}

int almpc_debug_poison_lds(almpc_handle* h) {
    if (!h) return ALMPC_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    const int bytes = 160 * 1024;
    HIP_TRY(h, ensure_dyn_lds(reinterpret_cast<const void*>(k_poison_lds), (size_t)(bytes)));
    // one workgroup owns a whole CU's LDS; several waves of workgroups so that every CU is visited
    hipLaunchKernelGGL(k_poison_lds, dim3(1024), dim3(1024), bytes, h->stream, 0x7ff8dead0000beefULL, bytes / 8,
                       reinterpret_cast<unsigned long long*>(h->dSglobal));
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return ALMPC_OK;
}

#ifdef ALMPC_STAMPS
// diagnostic build: allocate / fetch the stamp buffer ([waves][16] int64)
int almpc_dbg_stamps_enable(almpc_handle* h, int waves) {
    long long* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), (size_t)waves * 16 * sizeof(long long)) != hipSuccess) return -3;
    (void)hipMemset(d, 0, (size_t)waves * 16 * sizeof(long long));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(almpc::g_stamps), &d, sizeof(d));
    return 0;
}
int almpc_dbg_stamps_fetch(almpc_handle* h, long long* out, int waves) {
    long long* d = nullptr;
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(&d, HIP_SYMBOL(almpc::g_stamps), sizeof(d));
    if (!d) return -1;
    return hipMemcpy(out, d, (size_t)waves * 16 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#endif

}  // extern "C"

#include "almpc_hostio.inc.h"
