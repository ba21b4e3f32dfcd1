/*
 * almpc.h -- C ABI of libalmpc.so, the MI355X (gfx950) batched condensed-QP MPC solve engine.
 *
 * Drop-in boundary for ONE path of AutomationLabsModelPredictiveControl.jl (reference paths are
 * relative to /root/reference): the per-step solve that the reference runs through
 *   JuMP.Model(OSQP.Optimizer)             src/sub/solver_selection.jl:92-98   (solver slot)
 *   update_initialization!(C, x0)           src/main/computation_mpc.jl:17-29
 *   calculate!(C)                           src/main/computation_mpc.jl:38-55
 * on the QP posed by
 *   _model_predictive_control_modeler_implementation(::LinearProgramming, ...)
 *                                           src/sub/model_modeler_implementation/linear/mpc_modeler_implementation_linear.jl:20-103
 *   _create_weights_coefficients            src/sub/design_mpc.jl:264-283
 *   _create_terminal_ingredient             src/sub/design_mpc.jl:298-394
 *   _create_quadratic_cost_function         src/sub/design_mpc.jl:405-468
 * for a BATCH of independent controller instances that share (A, B, weights, box).
 *
 * Conventions
 *  - every matrix is column-major Float64, i.e. the memory of a Julia Matrix{Float64}
 *    (zero-copy `ccall` with Ptr{Float64}); batch is the slowest axis:
 *    x0[batch][n], u[batch][N][m], x[batch][N+1][n]  ==  Julia Array{Float64,3}(m, N, batch) etc.
 *  - the caller owns every host buffer; the library owns device memory behind the handle.
 *  - every entry point returns 0 (ALMPC_OK) or a negative almpc_status; nothing throws across
 *    the boundary; almpc_last_error(h) gives the message of the last failure on that handle.
 *  - per-instance solver outcome goes to status[] (almpc_solve_status), never to the return code.
 *  - a handle is bound to one HIP device and one stream; it is not thread-safe; distinct handles
 *    are independent.  One process per GPU: multi-GPU runs create one handle per rank on its own
 *    shard of the batch (instances never interact, so there is no data-path collective).
 *  - there is NO CPU fallback: if no gfx950 device is usable almpc_create fails with
 *    ALMPC_ERR_NO_DEVICE.
 */
#ifndef ALMPC_H
#define ALMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALMPC_VERSION_MAJOR 0
#define ALMPC_VERSION_MINOR 1

typedef struct almpc_handle almpc_handle;

typedef enum almpc_status {
    ALMPC_OK = 0,
    ALMPC_ERR_INVALID = -1,      /* bad argument (null pointer, non-positive size, ...)          */
    ALMPC_ERR_NO_DEVICE = -2,    /* no usable HIP device / device id out of range                */
    ALMPC_ERR_HIP = -3,          /* a HIP runtime call failed (message in almpc_last_error)      */
    ALMPC_ERR_UNSUPPORTED = -4,  /* size or option outside what the kernels are built for        */
    ALMPC_ERR_NOT_DESIGNED = -5, /* calculate/get before design                                  */
    ALMPC_ERR_NUMERIC = -6       /* design-time numerical failure (DARE / Cholesky)              */
} almpc_status;

/* per-instance outcome written by almpc_calculate */
typedef enum almpc_solve_status {
    ALMPC_SOLVED = 0,          /* KKT conditions hold: ADMM met its tolerance and/or the polish certified */
    ALMPC_MAX_ITER = 1,        /* ADMM hit max_iter and polish was off or hit its own cap                   */
    ALMPC_NON_FINITE = 2,      /* a non-finite value appeared (bad inputs)                                  */
    ALMPC_INFEASIBLE = 3       /* state rows only: no point satisfies the box / terminal equality           */
} almpc_solve_status;

/*
 * Solver options.  The ADMM fields are OSQP's settings of the same name (the reference builds
 * OSQP.Optimizer without attributes, src/sub/solver_selection.jl:94-95, i.e. OSQP defaults:
 * rho 0.1, sigma 1e-6, alpha 1.6, eps_abs = eps_rel = 1e-3, check_termination 25, max_iter 4000,
 * polish off).  Defaults here differ in two documented places: `polish` is ON (it is what brings
 * u* within 1e-5 of the exact optimum) and `max_iter` is 25, one check interval (ADMM only has to
 * identify the active set for the polish).  almpc_default_opts fills the defaults.
 */
typedef struct almpc_opts {
    double rho;
    double sigma;
    double alpha;
    double eps_abs;
    double eps_rel;
    int32_t max_iter;
    int32_t check_every;
    int32_t polish;           /* 0 off, 1 on                                                   */
    int32_t polish_max_iter;  /* cap on active-set changes per instance; <=0 -> 2*nz+50        */
    int32_t warm_start;       /* 1: start ADMM from the previous step's (x, z, y) of each instance */
    int32_t reserved[3];      /* reserved[0]: flag bits ALMPC_OPT_*; the others must be 0            */
} almpc_opts;

/* opts.reserved[0] bits */
/* The ADMM state (x, y) of this step is NOT kept for a warm start of the next one (a cold-start-every-step caller, e.g. a Monte-Carlo
 * sweep, saves their 2 * nz doubles of HBM writes per instance; the polish gets the signs of y as 4 bytes per 16 rows).  Results
 * are identical.  The next almpc_calculate on the handle must then not ask for warm_start (ALMPC_ERR_INVALID).  Shared-model
 * steps with polish = 1 only; ignored otherwise. */
#define ALMPC_OPT_NO_WARM_STATE 0x1

void almpc_default_opts(almpc_opts* opts);

/* flags for almpc_create */
#define ALMPC_FLAG_TIMING 0x1u /* record HIP events around every kernel of almpc_calculate */
/*
 * Structured (non-condensed) solve: the handle solves the multiple-shooting form the reference itself builds (states and inputs per
 * stage, dynamics as constraints: ..linear.jl:48-60) stage by stage -- no condensed Hessian, so no m*N <= 128 limit and no loss of
 * definiteness for open-loop unstable models over long horizons.  Solver: k_sdual, a dual active-set
 * method in constraint space whose Ghat columns are affine Riccati sweeps with the unconstrained feedback gains (the Riccati
 * recursion runs once per model, a working-set change costs two O(N (n^2 + n m)) sweeps): input box, STATE BOX on every stage
 * (..linear.jl:62-70), TERMINAL EQUALITY (src/sub/design_mpc.jl:330-331) and the INPUT-RATE WEIGHT S (src/sub/design_mpc.jl:423-446:
 * the stage state is then [e_k; v_{k-1}]); working sets up to 128 rows; infeasible instances are reported as ALMPC_INFEASIBLE.
 * Behind it, for an input box without S, k_riccati (primal active set with Riccati-recursion subproblems) takes the instances the dual
 * method leaves without a certificate (a saturated open-loop unstable plant: Ghat_WW numerically singular).
 * Limits: n + m <= 64 lanes (n <= 48 without S, n + m <= 48 with S; m <= 16), (N + 1)(n + m) doubles twice in 160 KB of LDS
 * (quadrotor: N <= 255); shapes outside them but inside n <= 32, m <= 16, m*N <= 1024 run k_riccati alone (input box only).
 * Entry points of such a handle: almpc_design_shared (xmin / xmax, almpc_set_terminal_equality, S honoured; rho / sigma unused),
 * almpc_design_batched (+ almpc_set_state_box), almpc_sqp_fnn_* (every QP of the loop in its stage-wise form), almpc_set_reference,
 * almpc_update_initialization(_device), almpc_calculate(_async) (opts.polish_max_iter caps the working-set changes; <= 0: 20 rows + 50),
 * almpc_get_results (iters: 0, polish_iters: scans + steps of the dual method), almpc_get_design (P only), almpc_comm_*.
 */
#define ALMPC_FLAG_STRUCTURED 0x2u

/*
 * Create a solver for `batch` instances of an (n states, m inputs, horizon N) controller on HIP
 * device `device_id`.  Supported: 1 <= m*N <= 128, 1 <= n <= 64 (ALMPC_FLAG_STRUCTURED: see there).
 */
int almpc_create(almpc_handle** out, int n, int m, int N, int batch, int device_id, uint32_t flags);
void almpc_destroy(almpc_handle* h);
const char* almpc_last_error(const almpc_handle* h);

/*
 * Design for a model shared by all instances (replaces _model_predictive_control_design for the
 * linear system, src/sub/design_mpc.jl:54-129).
 *   A n*n, B n*m, Q n*n, R m*m, S m*m (NULL -> 0), P n*n (NULL -> DARE(A,B,Q,R) as
 *   src/sub/design_mpc.jl:327), umin/umax m.  xmin/xmax n, or NULL: non-NULL adds the state box
 *   x_min <= x[:,k] <= x_max for k = 1..N+1, i.e. the reference's kw `mpc_state_constraint`
 *   (..linear.jl:62-70); needs n*N + m*N <= 512.  rho/sigma are fixed at design time because the
 *   shared KKT inverse depends on them.
 * Builds Phi, Gamma blocks, H = 2(Gamma' Qbar Gamma + Rbar + D'SbarD), F = 2 Gamma' Qbar Phi,
 * the Jacobi scaling, (H' + (sigma+rho) I)^-1 and H'^-1 on the device.
 */
/*
 * Terminal constraint e_x[:,N+1] == 0 (mpc_terminal_ingredient = "equality", src/sub/design_mpc.jl:330-331):
 * call with 1 BEFORE almpc_design_shared.  Problems with state rows (state box and/or terminal equality) are
 * finished by a dual active-set method in constraint space (k_polish_gen, working sets up to 32 rows; instances beyond that are
 * redone by k_polish_gen64, up to 64 rows; what is still undecided goes to the stage-wise redo of almpc_set_structured_fallback,
 * up to 128 rows); they require opts.polish = 1.
 */
int almpc_set_terminal_equality(almpc_handle* h, int on);

/*
 * ADMM penalty profile, call BEFORE almpc_design_shared.  0 (default): one scalar rho for every row, as OSQP does for
 * inequality rows.  1: stiffness-matched rho_i = rho / G_ii with G = H'^-1 (a diagonal preconditioning of the
 * constraint rows, the role OSQP's Ruiz scaling E plays): rows in soft directions of the Hessian get a small penalty.
 * On the benchmark plant this cuts the active-set work after 10 ADMM iterations by ~3x; `rho` of
 * almpc_design_shared is then the numerator (the benchmark uses 45 with 6 iterations: sweep in DESIGN.md section 4).
 */
int almpc_set_rho_profile(almpc_handle* h, int mode);

/*
 * Step fusion (default on): when the tile of 16 instances is 8 waves (nz in 113..128), polish = 1, no state rows and the
 * buffers fit LDS, a step is ONE kernel (ADMM phase, then the polish of the same 16 instances by the same workgroup).
 * 0 restores the two-kernel path (k_admm, k_polish), whose stage times the timing entry points can then separate.
 * Results are identical.  May be changed at any time.
 */
int almpc_set_step_fusion(almpc_handle* h, int on);

/*
 * Redo of the instances a condensed step leaves without a certificate in the multiple-shooting form, by the stage-wise solvers of
 * ALMPC_FLAG_STRUCTURED, starting from the step's own result.  Which instances, exactly:
 *   - status ALMPC_MAX_ITER (1: an active-set finish that ran into its cap or out of room -- in practice per-instance linearisations
 *     that are open-loop unstable, and state-row instances at the edge of feasibility) go to the dual solver k_sdual (input box,
 *     state box, terminal equality, S);
 *   - with an input box only and S = 0 the primal solver k_riccati then takes everything that is still not ALMPC_SOLVED, i.e. also
 *     status ALMPC_NON_FINITE (2) instances whose condensed problem was flagged indefinite to working precision (a design flag, not
 *     a non-finite input: the stage-wise form does not form that Hessian).  With state rows or S such an instance keeps status 2.
 * Instances that were solved are not touched; an instance the finish has already found infeasible keeps its verdict
 * (ALMPC_INFEASIBLE); an undecided state-row instance that turns out infeasible comes back as ALMPC_INFEASIBLE.
 *   DEFAULT: ON wherever those solvers cover the design (shape limits of ALMPC_FLAG_STRUCTURED), for every design of a condensed
 *   handle: shared (incl. state rows, terminal equality, S), per instance, re-linearisation pipeline, SQP loop (the QP of an iteration
 *   in its stage-wise form for the instances whose condensed Hessian came out indefinite, instead of skipping them).
 *     - the SQP loop: the redo kernels are enqueued inside every iteration.
 *     - every other condensed design (shared model -- the headline path, tens of microseconds per step --, per-instance models,
 *       re-linearisation pipeline; input box or state rows): LAZILY -- the finish counts the instances it leaves with ALMPC_MAX_ITER
 *       into a host-visible word and the redo runs at the next point where the host looks at results through a synchronous call
 *       (almpc_calculate, almpc_synchronize, almpc_get_results, almpc_get_first_input, almpc_relin_fnn_step): no launch on the
 *       step path (two idle redo launches cost 14 us -- of a 61 us step on the headline path; with state rows a stage-wise solve
 *       of ONE edge-of-feasibility instance takes about a millisecond, and 1 - 12 instances in 4096 need one).  The redo takes x0
 *       from the step's own result (x[:, 1]), so handing over the next x0 before looking at the results is safe; references and
 *       models must still be the step's.  A caller that reads results only through the asynchronous tickets, or consumes them on
 *       the device (almpc_advance_plant / almpc_relin_fnn_advance loops), asks for the eager form with on = 1.
 *   on = 1: required (a design the stage-wise solvers cannot serve is an error) and always eager;  on = 0: off.
 * Call BEFORE the design.
 */
int almpc_set_structured_fallback(almpc_handle* h, int on);

/*
 * Start of the NEXT almpc_calculate(_async) of a structured handle `h` from the inputs of `src`'s last step (any handle on the
 * same device with the same n, m, batch and a horizon <= h's; enqueued on h's stream behind src's step, no host wait): the stages
 * `src` has start at its inputs, the ones beyond at the input reference.  Horizon continuation: the active bounds of an MPC problem
 * with the DARE terminal weight sit in the early stages, so the condensed path at a short horizon (tens of microseconds) hands the
 * stage-wise path at a long one its working set -- the quadrotor at N = 50 from N = 30: at most 4 working-set changes instead of up
 * to 123 from the clipped LQR start.  The optimum does not depend on the start.  A structured handle also honours opts.warm_start = 1:
 * the start is then its own previous inputs shifted by one stage (the receding-horizon shift, last stage repeated).
 */
int almpc_set_start_from(almpc_handle* h, almpc_handle* src);

/*
 * State box of the designs that have no xmin / xmax arguments of their own -- almpc_design_batched, almpc_design_ltv, the
 * re-linearisation pipeline: x_min <= x[:,k] <= x_max for stages 1..N+1, the rows kw `mpc_state_constraint` adds in every delegate
 * of the reference (src/sub/model_modeler_implementation/linear/mpc_modeler_implementation_linear.jl:62-70,
 * .../fnn/mpc_modeler_implementation_fnn.jl:146-153).  Takes effect at the next such design; NULL, NULL removes it.  Together with
 * almpc_set_terminal_equality these designs then build one constraint-space matrix per instance (k_ghat_inst) and the step's exact
 * finish is the dual active-set kernel of the shared-model state rows with per-instance operands.  Time-varying designs and the SQP
 * loop: the box is on xbar + dx, the terminal equality reads xbar + dx = x_ref at stage N+1; an SQP iteration whose QP is infeasible
 * is skipped for that instance (almpc_sqp_fnn_skipped) -- there is no elastic mode.  Structured handles (ALMPC_FLAG_STRUCTURED) take the
 * same calls: there the rows are coordinates of the stage-wise trajectory (k_sdual), no constraint-space matrix is built.
 */
int almpc_set_state_box(almpc_handle* h, const double* xmin, const double* xmax);

int almpc_design_shared(almpc_handle* h, const double* A, const double* B, const double* Q,
                        const double* R, const double* S, const double* P, const double* umin,
                        const double* umax, const double* xmin, const double* xmax, double rho,
                        double sigma);

/*
 * Design with a model PER INSTANCE: A_batch [batch][n*n], B_batch [batch][n*m] (each block column-major).  This is the
 * linear-programming design of the reference applied to the linearisation of a black-box model at every instance's own
 * point (the reference linearises once, at the first reference, src/sub/model_modeler_implementation/fnn/
 * mpc_modeler_implementation_fnn.jl:38-46, and then builds the QP of ..linear.jl:48-100 from (A, B): given (A_i, B_i)
 * the QP of instance i is that QP; BASELINE.json configs[3] re-linearises every step, SURVEY.md section 8b
 * `almpc_design_batched`).  Q, R, S, umin, umax as in almpc_design_shared.  P: NULL -> DARE(A_i, B_i, Q, R) per instance
 * (host, src/sub/design_mpc.jl:327); else one n*n matrix (P_per_instance = 0) or [batch][n*n] (P_per_instance = 1).
 * Prediction matrices, condensed Hessian (FP64 MFMA contraction), scaling and both inverses are built on the device for
 * every instance; the step then runs one workgroup per instance with the instance's KKT inverse in LDS.  State rows:
 * almpc_set_state_box / almpc_set_terminal_equality before the design.  Not available with per-instance models:
 * almpc_advance_plant.  A later
 * almpc_design_shared switches the handle back to the shared-model path.
 */
int almpc_design_batched(almpc_handle* h, const double* A_batch, const double* B_batch, const double* Q,
                         const double* R, const double* S, const double* P, int P_per_instance,
                         const double* umin, const double* umax, double rho, double sigma);

/*
 * Design with TIME-VARYING models per instance: the QP of one SQP / multiple-shooting iteration around the trajectory
 * (xbar, ubar) of every instance (BASELINE.json configs[4]: SQP outer loop around the condensed-QP kernel; the reference has no
 * such path -- its NLP methods were removed, CHANGELOG.md:5 -- so only the QP itself has a reference meaning: with
 * A_k = A, B_k = B, c = 0 it is the QP of almpc_design_batched written in v = u - ubar).
 *   A_all [batch][N][n*n], B_all [batch][N][n*m]: Jacobians of x+ = f(x, u) at (xbar_k, ubar_k), k = 0..N-1 (column-major blocks);
 *   c_all [batch][N][n] or NULL: defects f(xbar_k, ubar_k) - xbar_{k+1};  xbar [batch][N+1][n] (xbar_0 = x0), ubar [batch][N][m];
 *   xref [N+1][n] / uref [N][m] or NULL (zeros): references of the tracking cost, shared;  Q, R, S, umin, umax as in
 *   almpc_design_shared;  P: terminal weight, n*n (P_per_instance = 0) or [batch][n*n] (required).
 * Linearised dynamics dx_{k+1} = A_k dx_k + B_k v_k + c_k, dx_0 = 0, cost of src/sub/design_mpc.jl:405-468 in (xbar + dx, ubar + v),
 * bounds umin <= ubar + v <= umax.  almpc_calculate then solves for v; almpc_get_results returns u = ubar + v and e_u = v
 * (x / e_x are not defined for an LTV design: roll the nonlinear model out on the caller's side).  almpc_set_reference is
 * not used with this design.  Needs nz^2 + 3 n nz + 4 n^2 + ... doubles <= 160 KB of LDS (nz <= ~120 for n = 12).
 */
int almpc_design_ltv(almpc_handle* h, const double* A_all, const double* B_all, const double* c_all, const double* xbar,
                     const double* ubar, const double* xref, const double* uref, const double* Q, const double* R,
                     const double* S, const double* P, int P_per_instance, const double* umin, const double* umax,
                     double rho, double sigma);

/* Gradient q (nz, unscaled, in v) of one instance's QP after almpc_design_ltv (parity hook; H: almpc_get_design_instance). */
int almpc_get_gradient_instance(almpc_handle* h, int instance, double* q);

/*
 * SQP outer loop for a black-box Fnn model, resident on the device (BASELINE configs[4]).  The problem is the reference's
 * NonLinearProgramming branch for Fnn models (.../fnn/mpc_modeler_implementation_fnn.jl:73-189): the quadratic cost of
 * _create_quadratic_cost_function (src/sub/design_mpc.jl:405-468) subject to x[:,k+1] = fnn(x[:,k], u[:,k]) and the input box,
 * which the reference hands to Ipopt (src/sub/solver_selection.jl:100-104).  Here: Gauss-Newton SQP with multiple shooting and
 * steps of length `step_scale`; every iteration = k_fnn_jacobian along the trajectory, the time-varying condensed QP of
 * almpc_design_ltv, the per-instance step kernels, and the trajectory update -- all on the handle's stream, no host round trip.
 * A fixed point (step and defects zero) is a KKT point of the NLP: the QP gradient is the exact NLP gradient.
 * The first iteration after `start` finds its working-set guess by ADMM, the following ones take it from the iterate (inputs
 * that sit on a bound) and skip the ADMM phase and its KKT inverse; the exact finish makes the two equivalent.
 *
 *   setup    network in the layout of almpc_fnn_linearize, shared references xref n*(N+1) / uref m*N (NULL: zeros), weights,
 *            P (required; shared or per instance), input box, ADMM rho/sigma.  Replaces any earlier design of the handle.
 *   start    x0 [batch][n]; u_guess [batch][N][m] or NULL (the input reference); both clipped to the box.  The state
 *            trajectory starts as the network's own rollout from x0 (zero defects).
 *   iterate  `iters` iterations with QP options `opts` (NULL: defaults).  step_inf[it] / defect_inf[it] (nullable, length
 *            iters): max over the batch of |v|_inf and of |f(xbar,ubar) - xbar+|_inf BEFORE iteration it's update.
 *            Afterwards almpc_get_results returns the iterate: x, u, e_x = x - xref, e_u = u - uref, status/iters of the last QP.
 *            An instance whose condensed Hessian is not positive definite to working precision (open-loop unstable
 *            linearisation over a long horizon) or whose QP solution is non-finite keeps its last good iterate; the call then
 *            returns ALMPC_ERR_NUMERIC after finishing all other instances; almpc_sqp_fnn_skipped tells which.
 */
int almpc_sqp_fnn_setup(almpc_handle* h, int H, int L, int activation, const double* W_in, const double* W_h, const double* b_h,
                        const double* W_out, const double* xref, const double* uref, const double* Q, const double* R,
                        const double* S, const double* P, int P_per_instance, const double* umin, const double* umax, double rho,
                        double sigma);
int almpc_sqp_fnn_start(almpc_handle* h, const double* x0, const double* u_guess);
int almpc_sqp_fnn_iterate(almpc_handle* h, int iters, double step_scale, const almpc_opts* opts, double* step_inf,
                          double* defect_inf);
int almpc_sqp_fnn_skipped(almpc_handle* h, int32_t* skipped /* [batch], 1 = some iteration was skipped */);
/* Step rule of almpc_sqp_fnn_iterate.  0 (default): every instance takes steps of length `step_scale` (full Gauss-Newton steps
 * are not globally convergent: they can end in a cycle when the tracking residual is large).  1: safeguarded by the l1 merit
 * function phi = J + mu |f(x,u) - x+|_1 (mu = 2 max(|P|, |Q|)), tested a posteriori with the network outputs the next
 * linearisation computes anyway: a step after which phi did not decrease is taken back and re-taken at half the length (that
 * iteration's QP is void for the instance), accepted steps double the factor back up to 1, and at 1/64 a step is accepted
 * regardless.  step_inf / defect_inf then cover the instances whose QP counted.  The factor restarts at 1 in `start`. */
int almpc_sqp_fnn_set_step_rule(almpc_handle* h, int rule);

/*
 * QP solver of the SQP loop (call before almpc_sqp_fnn_setup): 0 (default) the condensed path -- k_design_ltv, per-instance factors,
 * the step kernels --, 1 the stage-wise form for EVERY instance and iteration: k_sgains (Riccati recursion of the iteration's
 * unconstrained QP on (A_k, B_k, c_k)) + k_sdual, started from the iterate's own working set (v = 0).  No Hessian is formed or
 * inverted.  Same QP, same optimum (parity: oracle sqp_fnn(structured="dual")); takes the state box, the terminal equality and the
 * input-rate weight as the condensed route does.  Shape limits of ALMPC_FLAG_STRUCTURED.
 */
int almpc_sqp_fnn_set_structured(almpc_handle* h, int on);

/*
 * Per-step re-linearisation of a black-box Fnn model, resident on the device (BASELINE.json configs[3]).  The reference
 * linearises a black-box model ONCE, at the first reference (.../fnn/mpc_modeler_implementation_fnn.jl:38-46), and then runs the
 * linear path; this is the extension in which every instance is re-linearised at its own current state in every step: one call =
 * Jacobians A_i, B_i = d fnn / d(x, u) at (x0_i, u_ref[:,1]) written straight into the handle's per-instance model slots
 * (k_fnn_jacobian) -> the reference's QP for every (A_i, B_i) (the design kernels of almpc_design_batched) -> the step.  No host
 * pointer is touched: x0 comes from almpc_update_initialization(_device), results as after almpc_calculate.
 *   setup   network in the layout of almpc_fnn_linearize; xref n*(N+1), uref m*N shared (NULL: zeros); weights; P n*n (required:
 *           the reference takes the terminal weight from the linearisation at the LAST reference, src/sub/design_mpc.jl:312-327 --
 *           almpc_fnn_linearize + almpc_dare); input box; rho, sigma.  Replaces any earlier design of the handle.
 *   step    opts as almpc_calculate.  An instance whose condensed Hessian comes out without a positive diagonal / pivot (flagged by
 *           the design kernels) gets status ALMPC_NON_FINITE instead of a host-side error.
 *           opts->warm_start = 1 (after a solved step of this pipeline): the working-set guess is the previous step's input
 *           trajectory shifted by one stage instead of an ADMM phase, and the design needs one inverse per instance instead of two
 *           (the exact finish makes the two starts equivalent).
 *   advance x0 <- fnn(x0, u[:,1]) on the device: the closed loop of the black-box model itself, no host round trip.
 *   timing  (ALMPC_FLAG_TIMING) milliseconds of the last step's three stages.
 * On an ALMPC_FLAG_STRUCTURED handle (round 5; m N beyond the condensed limit -- the reference's delegation has no horizon limit,
 * .../fnn/mpc_modeler_implementation_fnn.jl:23-58) the same calls run the stage-wise route: Jacobians -> the Riccati recursion of
 * every instance's own unconstrained problem (k_sgains) -> the stage-wise dual active set (k_sdual) with the input box, the state box,
 * the terminal equality and S; rho / sigma are not used; opts->warm_start = 1 starts from the previous inputs shifted by one stage.
 * Shape limits of ALMPC_FLAG_STRUCTURED; an instance whose recursion meets a non-positive R + B'PB is left with ALMPC_MAX_ITER.
 */
int almpc_relin_fnn_setup(almpc_handle* h, int H, int L, int activation, const double* W_in, const double* W_h, const double* b_h,
                          const double* W_out, const double* xref, const double* uref, const double* Q, const double* R,
                          const double* S, const double* P, const double* umin, const double* umax, double rho, double sigma);
int almpc_relin_fnn_step(almpc_handle* h, const almpc_opts* opts);
int almpc_relin_fnn_step_async(almpc_handle* h, const almpc_opts* opts);
int almpc_relin_fnn_advance(almpc_handle* h);
int almpc_relin_fnn_timing(almpc_handle* h, float* ms_jacobian, float* ms_design, float* ms_step);

/* H (nz*nz), F (nz*n), d (nz) of one instance after almpc_design_batched (any pointer may be NULL). */
int almpc_get_design_instance(almpc_handle* h, int instance, double* H, double* F, double* d);

/*
 * References (replaces _design_reference_mpc, src/main/main_mpc.jl:105-117, and the JuMP.fix of
 * x_reference/u_reference, ...linear.jl:90-100).  per_instance = 0: xref n*(N+1), uref m*N shared
 * by all instances.  per_instance = 1: xref [batch][N+1][n], uref [batch][N][m].
 * Default after design: all zeros.
 */
int almpc_set_reference(almpc_handle* h, const double* xref, const double* uref, int per_instance);

/* x0: host pointer, [batch][n].  Replaces update_initialization! (src/main/computation_mpc.jl:17-29). */
int almpc_update_initialization(almpc_handle* h, const double* x0);
/* same, x0 already in device memory of h's device (HBM-resident benchmark / closed loop on device) */
int almpc_update_initialization_device(almpc_handle* h, const double* d_x0);

/*
 * One MPC step for every instance: gradient f = F e0, ADMM, polish, rollout.  Replaces
 * JuMP.optimize! + the four JuMP.value reads of calculate! (src/main/computation_mpc.jl:41-53).
 * opts == NULL -> defaults.  opts->rho / opts->sigma must equal the design values (or be 0).
 * Synchronous: results are complete in device memory on return.
 */
int almpc_calculate(almpc_handle* h, const almpc_opts* opts);
/* enqueue only (no host sync); pair with almpc_synchronize */
int almpc_calculate_async(almpc_handle* h, const almpc_opts* opts);
int almpc_synchronize(almpc_handle* h);

/*
 * Closed loop without host round trips: x0 <- A x0 + B u[:,1] on the device, per instance, with the (A, B) of the design
 * and the first input of the last almpc_calculate (enqueued on the handle's stream; no host sync).  Follow with
 * almpc_calculate(_async) -- typically with opts.warm_start = 1 -- for the next receding-horizon step.
 */
int almpc_advance_plant(almpc_handle* h);

/*
 * Copy results to host.  Any pointer may be NULL.  Layouts: x, e_x [batch][N+1][n];
 * u, e_u [batch][N][m]  (ModelPredictiveControlResults, src/types/types.jl:134-139);
 * status, iters, polish_iters [batch].
 */
int almpc_get_results(almpc_handle* h, double* x, double* e_x, double* u, double* e_u,
                      int32_t* status, int32_t* iters, int32_t* polish_iters);

/*
 * Host-facing step path.  The reference's per-step contract is host in / host out (update_initialization!(C, x0),
 * src/main/computation_mpc.jl:17-29; calculate! leaves x, e_x, u, e_u in host matrices, src/main/computation_mpc.jl:50-53), and a
 * receding-horizon caller applies u[:,1] only.  These entry points move exactly what is asked for through pinned staging owned by
 * the handle, on copy streams of their own, so that the transfers of one step run under the kernel of the next:
 *   almpc_update_initialization_async   x0 [batch][n] (host): copied into a pinned slot of the handle, which the kernels of the next
 *                                       almpc_calculate(_async) read in place over the link (393 KB at the benchmark shape, requested
 *                                       under the operands every workgroup loads anyway): no HIP call, no device-side copy.  Returns
 *                                       at once; blocks only if both slots are still being read (two steps in flight).
 *   almpc_x0_staging                    zero-copy variant of the above: *x0_slot = the pinned host buffer ([batch][n]) that the NEXT
 *                                       almpc_update_initialization_async will use.  Write the states there (e.g. straight from the
 *                                       plant simulation) and pass the same pointer to almpc_update_initialization_async: the
 *                                       staging copy is skipped.  The pointer is valid until that call.
 *   almpc_get_results_async             asks for the results of the LAST enqueued step: `want` = mask of ALMPC_WANT_*.  The small
 *                                       ones (first inputs, status, iteration counts) are written by one pack kernel straight into
 *                                       the pinned slot, so the next step may start right behind it; x / e_x / u / e_u go out on a
 *                                       copy stream straight from the result buffers (the next step waits for that read-back).
 *                                       Returns a ticket >= 0 (or a negative almpc_status); the last 2 tickets are kept.
 *   almpc_get_results_wait              blocks until the ticket's read-back has landed, then copies the non-NULL arrays out of the
 *                                       pinned slot (layouts of almpc_get_results; u0 [batch][m] = u[:,1] of every instance).
 *   almpc_host_results                  zero-copy: pointers into the pinned slot of a ticket (NULL for arrays that were not asked
 *                                       for); valid after almpc_get_results_wait(h, ticket, NULL...) until 2 more requests.
 *   almpc_get_first_input               synchronous convenience: u0 [batch][m] of the last step (131 KB at the benchmark shape
 *                                       instead of the 32 MB of almpc_get_results).
 */
#define ALMPC_WANT_X 0x01u
#define ALMPC_WANT_E_X 0x02u
#define ALMPC_WANT_U 0x04u
#define ALMPC_WANT_E_U 0x08u
#define ALMPC_WANT_STATUS 0x10u
#define ALMPC_WANT_ITERS 0x20u
#define ALMPC_WANT_POLISH_ITERS 0x40u
#define ALMPC_WANT_FIRST_INPUT 0x80u
#define ALMPC_WANT_ALL 0xFFu
int almpc_update_initialization_async(almpc_handle* h, const double* x0);
int almpc_x0_staging(almpc_handle* h, double** x0_slot);
int almpc_get_results_async(almpc_handle* h, uint32_t want);
int almpc_get_results_wait(almpc_handle* h, int ticket, double* x, double* e_x, double* u, double* e_u, double* u0,
                           int32_t* status, int32_t* iters, int32_t* polish_iters);
int almpc_host_results(almpc_handle* h, int ticket, const double** x, const double** e_x, const double** u, const double** e_u,
                       const double** u0, const int32_t** status, const int32_t** iters, const int32_t** polish_iters);
int almpc_get_first_input(almpc_handle* h, double* u0);

/*
 * One process, several GPUs (almpc_create(..., n_devices, device_ids, ...); the reference API is one
 * process, one call: proceed_controller, src/main/main_mpc.jl:22-53).  A group is one handle per entry of device_ids (a device may
 * appear more than once), handle i on the contiguous shard [first_i, first_i + count_i) of the batch (sizes differ by at most
 * one); instances never interact, so a group call fans out over the handles and nothing on the step path synchronises across
 * devices: almpc_group_calculate_async enqueues on every device before almpc_group_synchronize waits for any.  Host arrays are
 * those of the single-handle calls with `batch` = the whole batch; per-instance arrays (models, terminal weights, guesses) are cut
 * along the shards.  Everything a handle can do has a group form: the options that are set before a design (almpc_group_set_*), the
 * shared and per-instance designs, the re-linearisation pipeline, the SQP loop, the host-facing read-back (tickets, zero-copy x0
 * slots); the synchronous design entry points run on one host thread per device, so that all devices design at the same time.
 * flags (ALMPC_FLAG_TIMING, ALMPC_FLAG_STRUCTURED) go to every handle; almpc_group_handle(g, i) reaches one handle directly.
 */
typedef struct almpc_group almpc_group;
int almpc_group_create(almpc_group** out, int n, int m, int N, int batch, int n_devices, const int* device_ids, uint32_t flags);
void almpc_group_destroy(almpc_group* g);
const char* almpc_group_last_error(const almpc_group* g);
int almpc_group_size(const almpc_group* g);
almpc_handle* almpc_group_handle(almpc_group* g, int i);
int almpc_group_shard(const almpc_group* g, int i, int* first, int* count);
int almpc_group_design_shared(almpc_group* g, const double* A, const double* B, const double* Q, const double* R, const double* S,
                              const double* P, const double* umin, const double* umax, const double* xmin, const double* xmax,
                              double rho, double sigma);
int almpc_group_set_reference(almpc_group* g, const double* xref, const double* uref, int per_instance);
int almpc_group_update_initialization(almpc_group* g, const double* x0);
int almpc_group_calculate(almpc_group* g, const almpc_opts* opts);
int almpc_group_calculate_async(almpc_group* g, const almpc_opts* opts);
int almpc_group_synchronize(almpc_group* g);
int almpc_group_get_results(almpc_group* g, double* x, double* e_x, double* u, double* e_u, double* u0, int32_t* status,
                            int32_t* iters, int32_t* polish_iters);
/* options of every handle (each takes effect at the next design, as the single-handle call says) */
int almpc_group_set_terminal_equality(almpc_group* g, int on);
int almpc_group_set_rho_profile(almpc_group* g, int mode);
int almpc_group_set_structured_fallback(almpc_group* g, int on);
int almpc_group_set_state_box(almpc_group* g, const double* xmin, const double* xmax);
/* almpc_design_batched: A_batch [batch][n*n], B_batch [batch][n*m], P NULL | n*n | [batch][n*n] (P_per_instance = 1) */
int almpc_group_design_batched(almpc_group* g, const double* A_batch, const double* B_batch, const double* Q, const double* R,
                               const double* S, const double* P, int P_per_instance, const double* umin, const double* umax, double rho,
                               double sigma);
/* almpc_relin_fnn_* (BASELINE configs[3]): the same network and references on every device; step_async + almpc_group_synchronize */
int almpc_group_relin_fnn_setup(almpc_group* g, int H, int L, int activation, const double* W_in, const double* W_h, const double* b_h,
                                const double* W_out, const double* xref, const double* uref, const double* Q, const double* R,
                                const double* S, const double* P, const double* umin, const double* umax, double rho, double sigma);
int almpc_group_relin_fnn_step(almpc_group* g, const almpc_opts* opts);
int almpc_group_relin_fnn_step_async(almpc_group* g, const almpc_opts* opts);
int almpc_group_relin_fnn_advance(almpc_group* g);
int almpc_group_advance_plant(almpc_group* g);
/* almpc_sqp_fnn_* (BASELINE configs[4]): x0 [batch][n], u_guess [batch][N][m] or NULL, P n*n or [batch][n*n]; iterate runs the devices'
 * loops at the same time, step_inf / defect_inf are maxima over the whole batch; skipped [batch] */
int almpc_group_sqp_fnn_set_structured(almpc_group* g, int on);
int almpc_group_sqp_fnn_set_step_rule(almpc_group* g, int rule);
int almpc_group_sqp_fnn_setup(almpc_group* g, int H, int L, int activation, const double* W_in, const double* W_h, const double* b_h,
                              const double* W_out, const double* xref, const double* uref, const double* Q, const double* R,
                              const double* S, const double* P, int P_per_instance, const double* umin, const double* umax, double rho,
                              double sigma);
int almpc_group_sqp_fnn_start(almpc_group* g, const double* x0, const double* u_guess);
int almpc_group_sqp_fnn_iterate(almpc_group* g, int iters, double step_scale, const almpc_opts* opts, double* step_inf,
                                double* defect_inf);
int almpc_group_sqp_fnn_skipped(almpc_group* g, int32_t* skipped);
/* host-facing path of the group: slots[i] = handle i's pinned x0 buffer ([count_i][n], almpc_x0_staging) -> write the shard's states
 * there -> almpc_group_update_initialization_staged(g, slots) (no copy);  read-back by ticket as almpc_get_results_async / _wait, the
 * arrays of _wait being those of almpc_group_get_results */
int almpc_group_x0_staging(almpc_group* g, double** slots /* [almpc_group_size(g)] */);
int almpc_group_update_initialization_staged(almpc_group* g, double* const* slots);
int almpc_group_get_results_async(almpc_group* g, uint32_t want);
int almpc_group_get_results_wait(almpc_group* g, int ticket, double* x, double* e_x, double* u, double* e_u, double* u0, int32_t* status,
                                 int32_t* iters, int32_t* polish_iters);

/* Design data for parity tests: H nz*nz, F nz*n (both unscaled, column-major), P n*n, d nz. */
int almpc_get_design(almpc_handle* h, double* H, double* F, double* P, double* d);

/*
 * Device pointers of the handle's result buffers (same layouts as almpc_get_results), for callers
 * that keep the loop on the GPU.  Valid until almpc_destroy.
 */
int almpc_device_results(almpc_handle* h, const double** d_x, const double** d_e_x,
                         const double** d_u, const double** d_e_u);

/*
 * Timing of the last almpc_calculate (needs ALMPC_FLAG_TIMING): milliseconds spent in the ADMM
 * kernel, the polish kernel(s), the rollout kernel, and the whole enqueue->done span, measured
 * with HIP events on the handle's stream.
 */
int almpc_get_timing(almpc_handle* h, float* ms_admm, float* ms_polish, float* ms_rollout,
                     float* ms_total);
/*
 * Timing over many steps: with ALMPC_FLAG_TIMING every almpc_calculate records its own set of events.
 * almpc_timing_reset forgets the recorded steps (and pre-creates events for `reserve_steps` steps so that
 * no event is created inside a timed region); almpc_timing_summary synchronises and returns the number of
 * steps recorded since the reset and the SUM of their per-stage milliseconds.
 */
int almpc_timing_reset(almpc_handle* h, int reserve_steps);
/* Record the events only on every `every`-th almpc_calculate (default 1).  One step's four event records cost ~14 us of
 * stream time on MI355X, so a throughput run samples (bench.py: every 16th step) instead of timing every step. */
int almpc_timing_set_stride(almpc_handle* h, int every);
int almpc_timing_summary(almpc_handle* h, int* steps, double* ms_admm, double* ms_polish,
                         double* ms_rollout, double* ms_total);
/* The individual samples behind almpc_timing_summary: per recorded step its four stage times (arrays of `cap` floats, any may be
 * NULL; the first min(cap, *count) entries are filled); *count = steps recorded since the reset.  Synchronises the stream. */
int almpc_timing_samples(almpc_handle* h, int cap, int* count, float* ms_admm, float* ms_polish, float* ms_rollout,
                         float* ms_total);

/*
 * Discrete algebraic Riccati solution P of  A'PA - P - A'PB (R + B'PB)^-1 B'PA + Q = 0  (host, n x n): what
 * ControlSystems.are(Discrete, A, B, Q, R) returns at src/sub/design_mpc.jl:327.  Exposed because a black-box model takes
 * its terminal weight from the linearisation at the LAST reference while its dynamics come from the FIRST
 * (src/sub/design_mpc.jl:312-327 vs .../fnn/mpc_modeler_implementation_fnn.jl:38-46): pass the result as P to the design.
 */
int almpc_dare(int n, int m, const double* A, const double* B, const double* Q, const double* R, double* P);

/*
 * Batched linearisation of a black-box Fnn model on the GPU: for each of `batch` points (x_i, u_i) the Jacobians
 * A_i = df/dx (n x n), B_i = df/du (n x m), both column-major, and optionally f_i = f(x_i, u_i).  Stands in for
 * AutomationLabsSystems.proceed_system_linearization (.../fnn/mpc_modeler_implementation_fnn.jl:42-46).  Layout as the
 * reference reads it from Flux.params (.../fnn/...:88-107): W_in H x (n+m) without bias or activation, L hidden layers
 * (W_h[l] H x H, b_h[l] H) with `activation` (0 identity, 1 relu, 2 tanh, 3 sigmoid, 4 swish; the
 * reference takes whatever NNlib function sits at f[2][1].sigma, src/sub/design_mpc.jl:472-483), W_out n x H without bias; all column-major.
 * Host pointers; synchronous.
 */
int almpc_fnn_linearize(int device_id, int n, int m, int H, int L, int activation, const double* W_in,
                        const double* W_h, const double* b_h, const double* W_out, int batch, const double* x,
                        const double* u, double* A, double* B, double* f);

/*
 * Multi-GPU: one process per GPU, each with its own handle on its contiguous shard of the batch; instances never interact, so no
 * collective is on the data path of a step.  RCCL (over xGMI) runs inside the library (dlopen("librccl.so") at first use) for what a
 * multi-GPU caller needs between steps:
 *   almpc_comm_unique_id   rank 0 makes the 128-byte id and hands it to the other ranks by any channel (file, socket, MPI, ...)
 *   almpc_comm_init        every rank, collectively: ncclCommInitRank on the handle's device
 *   almpc_comm_summary     out4 = {ranks in the job, instances with status != 0 over all ranks (sums); max ADMM iterations, max
 *                          polish iterations over all ranks (max)} of the last step; collective; synchronises the handle's stream
 *   almpc_comm_allgather_first_input   u[:,1] of every instance of every rank: [world][batch][m] doubles (32 B per quadrotor instance,
 *                          never the full trajectories); u0_all: host buffer or NULL; d_u0_all: receives the device pointer or NULL;
 *                          collective, on the handle's stream (synchronous only when u0_all is given)
 * ALMPC_ERR_UNSUPPORTED: librccl could not be loaded.
 */
#define ALMPC_COMM_ID_BYTES 128
int almpc_comm_unique_id(char* id128);
int almpc_comm_init(almpc_handle* h, const char* id128, int rank, int world);
int almpc_comm_summary(almpc_handle* h, int64_t* out4);
int almpc_comm_allgather_first_input(almpc_handle* h, double* u0_all, const double** d_u0_all);

/*
 * Test hook: overwrite the LDS of every compute unit with NaN bit patterns (a kernel that reads LDS it has not written
 * then produces NaNs instead of passing on stale values).  Synchronous.  Not needed by callers.
 */
int almpc_debug_poison_lds(almpc_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* ALMPC_H */
