# AlmpcHIP.jl -- thin Julia shim over libalmpc.so (include/almpc.h).
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no `julia`.  Every behaviour of the ABI is exercised
# through the same entry points from Python (tests/test_gpu_parity.py) and from C (tests/capi_harness.c, which passes
# column-major arrays exactly as `ccall` does); this file is the binding a maintainer of
# AutomationLabsModelPredictiveControl.jl would add, and julia/reference_hip.patch wires it into the reference.  It keeps the reference's names and call shapes
# (src/main/main_mpc.jl:22-53, src/sub/design_mpc.jl:54-129, src/main/computation_mpc.jl:17-55) and adds the
# solver tag "hip" next to osqp/scip/ipopt/auto (src/sub/solver_selection.jl:9-14).
module AlmpcHIP

export hip_solver_def, HipModeler, AlmpcOpts, design_hip, terminal_weight, set_state_rows!, design_batched!, design_sqp_fnn!, sqp_start!, sqp_iterate!,
       design_relin_fnn!, relin_step!, relin_advance!, update_initialization!, calculate!, read_results!,
       _model_predictive_control_computation, comm_unique_id, comm_init!, comm_summary, comm_allgather_first_input,
       calculate_async!, synchronize!, relin_step_async!, advance_plant!, start_from!, update_initialization_device!, device_results, set_step_fusion!,
       set_structured_fallback!, sqp_skipped, design_instance, gradient_instance, design_ltv!, fnn_linearize, dare, default_opts,
       get_timing, timing_reset!, timing_set_stride!, timing_summary, timing_samples, relin_timing, debug_poison_lds!,
       update_initialization_async!, x0_staging, results_async, results_wait!, host_results, first_input, first_input!, ALLOW_UNSOLVED,
       HipGroup, group_design_hip, group_handle, group_shard, group_update_initialization!, group_calculate!, group_calculate_async!,
       group_synchronize!, group_read_results!, group_set_state_rows!, group_set_rho_profile!, group_set_structured_fallback!,
       group_design_batched!, group_design_relin_fnn!, group_relin_step!, group_relin_advance!, group_advance_plant!, group_design_sqp_fnn!,
       group_sqp_start!, group_sqp_iterate!, group_sqp_skipped, group_x0_staging, group_update_initialization_staged!, group_results_async,
       group_results_wait!

const libalmpc = get(ENV, "ALMPC_LIB", "libalmpc.so")

struct hip_solver_def end            # new tag, to be made <: AbstractSolvers in src/types/types.jl:162-192

Base.@kwdef struct AlmpcOpts         # mirrors `almpc_opts` (72 bytes)
    rho::Cdouble = 0.1
    sigma::Cdouble = 1e-6
    alpha::Cdouble = 1.6
    eps_abs::Cdouble = 1e-3
    eps_rel::Cdouble = 1e-3
    max_iter::Int32 = 25
    check_every::Int32 = 25
    polish::Int32 = 1
    polish_max_iter::Int32 = 0
    warm_start::Int32 = 0
    reserved::NTuple{3,Int32} = (0, 0, 0)
end

mutable struct HipModeler            # what sits in tuning.modeler (`modeler::Any`, src/types/types.jl:115)
    handle::Ptr{Cvoid}
    n::Int; m::Int; N::Int; batch::Int
    opts::AlmpcOpts
    # what `calculate!` runs: :linear (almpc_calculate on the design in place), :relin (per-step re-linearisation of a black-box
    # model, almpc_relin_fnn_step), :sqp (the NonLinearProgramming branch: almpc_sqp_fnn_start from the last x0 + `sqp_iterations`)
    mode::Symbol
    sqp_iterations::Int
    x0::Vector{Float64}              # :sqp only: the last initialisation (the loop restarts from it in every calculate!)
end
HipModeler(h, n, m, N, batch, opts) = HipModeler(h, n, m, N, batch, opts, :linear, 20, Float64[])

function check(h, rc)
    rc == 0 && return
    msg = unsafe_string(ccall((:almpc_last_error, libalmpc), Cstring, (Ptr{Cvoid},), h))
    error("libalmpc error $rc: $msg")   # the reference throws from JuMP.value when no solution exists
end

"""
    design_hip(A, B, Q, R, S, umin, umax, N; batch = 1, device = 0, x_ref, u_ref, opts,
               xmin = nothing, xmax = nothing, terminal = "none", rho_profile = "scalar", P = nothing)

Design for `ConstrainedLinearControlDiscreteSystem` (replaces the JuMP model built at
src/sub/model_modeler_implementation/linear/mpc_modeler_implementation_linear.jl:20-103 and the objective of
src/sub/design_mpc.jl:405-468).  `P = nothing`: computed inside (DARE, src/sub/design_mpc.jl:327).
`xmin`/`xmax`: the state box of kw `mpc_state_constraint` (..linear.jl:62-70, stages 1..N+1).  `terminal = "equality"`:
`e_x[:, end] .== 0` (src/sub/design_mpc.jl:330-331); "none" / "neighborhood" add nothing, as in the reference; "contractive" is a
quadratic constraint and is refused.  `rho_profile`: "scalar" (OSQP) or "stiffness" (`almpc_set_rho_profile`).
"""
function design_hip(A::Matrix{Float64}, B::Matrix{Float64}, Q::Matrix{Float64}, R::Matrix{Float64},
                    S::Matrix{Float64}, umin::Vector{Float64}, umax::Vector{Float64}, N::Int;
                    batch::Int = 1, device::Int = 0, x_ref::Matrix{Float64}, u_ref::Matrix{Float64},
                    opts::AlmpcOpts = AlmpcOpts(), xmin::Union{Nothing,Vector{Float64}} = nothing,
                    xmax::Union{Nothing,Vector{Float64}} = nothing, terminal::String = "none",
                    rho_profile::String = "scalar", P::Union{Nothing,Matrix{Float64}} = nothing)
    n, m = size(B)
    terminal == "contractive" && error("terminal ingredient \"contractive\" is a quadratic constraint (src/sub/design_mpc.jl:333-340), not a QP row")
    (xmin === nothing) == (xmax === nothing) || error("give both xmin and xmax or neither")
    href = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:almpc_create, libalmpc), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Cint, Cint, Cint, UInt32),
               href, n, m, N, batch, device, 0)
    rc == 0 || error("almpc_create failed ($rc): no gfx950 device? (there is no CPU fallback)")
    h = href[]
    mod = HipModeler(h, n, m, N, batch, opts)
    finalizer(x -> ccall((:almpc_destroy, libalmpc), Cvoid, (Ptr{Cvoid},), x.handle), mod)
    check(h, ccall((:almpc_set_terminal_equality, libalmpc), Cint, (Ptr{Cvoid}, Cint), h, terminal == "equality" ? 1 : 0))
    check(h, ccall((:almpc_set_rho_profile, libalmpc), Cint, (Ptr{Cvoid}, Cint), h, rho_profile == "stiffness" ? 1 : 0))
    pP = P === nothing ? Ptr{Float64}(C_NULL) : pointer(P)
    pxmin = xmin === nothing ? Ptr{Float64}(C_NULL) : pointer(xmin)
    pxmax = xmax === nothing ? Ptr{Float64}(C_NULL) : pointer(xmax)
    GC.@preserve P xmin xmax check(h, ccall((:almpc_design_shared, libalmpc), Cint,
                   (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble),
                   h, A, B, Q, R, S, pP, umin, umax, pxmin, pxmax, opts.rho, opts.sigma))
    # references: n x (N+1) and m x N Julia matrices are already the ABI's [N+1][n] / [N][m] memory
    check(h, ccall((:almpc_set_reference, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cint),
                   h, x_ref, u_ref, 0))
    return mod
end

"""
    set_state_rows!(mod; xmin = nothing, xmax = nothing, terminal = "none")

State box (kw `mpc_state_constraint`; .../fnn/mpc_modeler_implementation_fnn.jl:52-58,146-153) and terminal equality
(src/sub/design_mpc.jl:330-331) of the designs without `xmin`/`xmax` arguments: call before `design_batched!`, `design_sqp_fnn!`,
`design_relin_fnn!`.
"""
function set_state_rows!(mod::HipModeler; xmin::Union{Nothing,Vector{Float64}} = nothing, xmax::Union{Nothing,Vector{Float64}} = nothing,
                         terminal::String = "none")
    (xmin === nothing) == (xmax === nothing) || error("give both xmin and xmax or neither")
    check(mod.handle, ccall((:almpc_set_terminal_equality, libalmpc), Cint, (Ptr{Cvoid}, Cint), mod.handle, terminal == "equality" ? 1 : 0))
    pmin = xmin === nothing ? Ptr{Float64}(C_NULL) : pointer(xmin)
    pmax = xmax === nothing ? Ptr{Float64}(C_NULL) : pointer(xmax)
    GC.@preserve xmin xmax check(mod.handle, ccall((:almpc_set_state_box, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}),
                                                   mod.handle, pmin, pmax))
    return mod
end

"terminal weight P (n x n) of the current design: what `_create_terminal_ingredient` returns as `P_cost` (src/sub/design_mpc.jl:327,393)"
function terminal_weight(mod::HipModeler)
    P = Matrix{Float64}(undef, mod.n, mod.n)
    check(mod.handle, ccall((:almpc_get_design, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                            mod.handle, C_NULL, C_NULL, P, C_NULL))
    return P
end

"""
    design_batched!(mod, A_batch, B_batch, Q, R, S, P, umin, umax; x_ref, u_ref)

One model per instance (`A_batch` n x n x batch, `B_batch` n x m x batch: Julia's column-major 3-arrays are the ABI's
[batch][n*n] / [batch][n*m] blocks): the linear-programming design of the reference for the linearisation of a black-box model at
every instance's own point (src/sub/model_modeler_implementation/fnn/mpc_modeler_implementation_fnn.jl:38-46 linearises once; the
per-step re-linearisation is BASELINE configs[3]).  `P` = nothing (DARE per instance), an n x n matrix, or n x n x batch.
"""
function design_batched!(mod::HipModeler, A_batch::Array{Float64,3}, B_batch::Array{Float64,3}, Q::Matrix{Float64},
                         R::Matrix{Float64}, S::Matrix{Float64}, P, umin::Vector{Float64}, umax::Vector{Float64};
                         x_ref::Matrix{Float64}, u_ref::Matrix{Float64})
    size(A_batch, 3) == mod.batch && size(B_batch, 3) == mod.batch || throw(DimensionMismatch("one model per instance"))
    pptr = P === nothing ? Ptr{Float64}(C_NULL) : pointer(P)
    pinst = (P !== nothing && ndims(P) == 3) ? 1 : 0
    GC.@preserve P check(mod.handle, ccall((:almpc_design_batched, libalmpc), Cint,
                   (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint,
                    Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble),
                   mod.handle, A_batch, B_batch, Q, R, S, pptr, pinst, umin, umax, mod.opts.rho, mod.opts.sigma))
    check(mod.handle, ccall((:almpc_set_reference, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cint),
                            mod.handle, x_ref, u_ref, 0))
    return mod
end

"""
    design_sqp_fnn!(mod, W_in, W_h, b_h, W_out, activation, Q, R, S, P, umin, umax; x_ref, u_ref)

NonLinearProgramming branch for an Fnn model (src/sub/model_modeler_implementation/fnn/mpc_modeler_implementation_fnn.jl:73-189,
which the reference solves with Ipopt): the same NLP by Gauss-Newton SQP on the device.  `W_in` H x (n+m), `W_h` H x H x L,
`b_h` H x L, `W_out` n x H as read from Flux.params (:88-107); `activation` 0 identity, 1 relu, 2 tanh, 3 sigmoid, 4 swish.
Then per step:  `sqp_start!(mod, x0)`;  `sqp_iterate!(mod, iters)`;  results through `calculate!`'s readers (`almpc_get_results`).
"""
function design_sqp_fnn!(mod::HipModeler, W_in::Matrix{Float64}, W_h::Array{Float64,3}, b_h::Matrix{Float64}, W_out::Matrix{Float64},
                         activation::Integer, Q::Matrix{Float64}, R::Matrix{Float64}, S::Matrix{Float64}, P::Matrix{Float64},
                         umin::Vector{Float64}, umax::Vector{Float64}; x_ref::Matrix{Float64}, u_ref::Matrix{Float64},
                         structured_qp::Bool = false)
    # structured_qp: every iteration's QP in the multiple-shooting form (k_riccati) instead of the condensed one
    check(mod.handle, ccall((:almpc_sqp_fnn_set_structured, libalmpc), Cint, (Ptr{Cvoid}, Cint), mod.handle, structured_qp ? 1 : 0))
    check(mod.handle, ccall((:almpc_sqp_fnn_setup, libalmpc), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble),
                   mod.handle, size(W_in, 1), size(W_h, 3), activation, W_in, W_h, b_h, W_out, x_ref, u_ref, Q, R, S, P, 0, umin, umax,
                   mod.opts.rho, mod.opts.sigma))
    return mod
end

sqp_start!(mod::HipModeler, x0::VecOrMat{Float64}) =
    check(mod.handle, ccall((:almpc_sqp_fnn_start, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), mod.handle, x0, C_NULL))

function sqp_iterate!(mod::HipModeler, iters::Integer; step::Float64 = 1.0, merit_safeguard::Bool = true)
    check(mod.handle, ccall((:almpc_sqp_fnn_set_step_rule, libalmpc), Cint, (Ptr{Cvoid}, Cint), mod.handle, merit_safeguard ? 1 : 0))
    st, de = zeros(iters), zeros(iters)
    o = Ref(mod.opts)
    check(mod.handle, ccall((:almpc_sqp_fnn_iterate, libalmpc), Cint, (Ptr{Cvoid}, Cint, Cdouble, Ref{AlmpcOpts}, Ptr{Float64}, Ptr{Float64}),
                            mod.handle, iters, step, o, st, de))
    return st, de
end

# update_initialization!(C, x0): x0 is a Vector (batch 1) or an n x batch Matrix (src/main/computation_mpc.jl:17-29)
function update_initialization!(mod::HipModeler, x0::VecOrMat{Float64})
    length(x0) == mod.n * mod.batch || throw(DimensionMismatch("x0 must hold n x batch values"))
    if mod.mode === :sqp
        mod.x0 = vec(copy(x0))      # the SQP loop uploads it in sqp_start! (almpc_sqp_fnn_start)
        return
    end
    check(mod.handle, ccall((:almpc_update_initialization, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float64}), mod.handle, x0))
end

# The library writes batch * n * (N+1) / batch * m * N doubles through these pointers: anything shorter would be a heap overflow
# (the reference allocates ONE instance's matrices, src/sub/design_mpc.jl:515-529; a batched caller must allocate the batch).
function check_result_sizes(mod::HipModeler, x, e_x, u, e_u)
    nx, nu = mod.n * (mod.N + 1) * mod.batch, mod.m * mod.N * mod.batch
    for (name, a, want) in (("x", x, nx), ("e_x", e_x, nx), ("u", u, nu), ("e_u", e_u, nu))
        a === nothing && continue
        length(a) == want || throw(DimensionMismatch("$name holds $(length(a)) values, the handle writes $want (n or m x horizon x batch = $(mod.batch))"))
    end
end

# What a per-instance status means to a caller of the reference's API.  The reference never checks the solver status and lets
# JuMP.value throw when there is no primal (src/main/computation_mpc.jl:41-53): it returns a solution or throws.  Here:
#   2 (non-finite) and 3 (infeasible: state box / terminal equality)  -> error, as JuMP.value would;
#   1 (no certificate: an iteration cap or a working set beyond 128 rows, after the stage-wise redo that libalmpc runs by default,
#      include/almpc.h almpc_set_structured_fallback)                  -> error as well, unless ALLOW_UNSOLVED[] is set: the arrays then hold
#      the best iterate (inputs inside their box, trajectory consistent with them) and the status vector says which instances.
#      With the exact finish switched off (opts.polish = 0) status 1 is OSQP's ITERATION_LIMIT, with which JuMP.value still returns: no error.
const ALLOW_UNSOLVED = Ref(false)
function throw_on_status(status; strict::Bool = true)
    any(==(2), status) && error("calculate!: non-finite values in at least one instance (no solution to read)")
    any(==(3), status) && error("calculate!: infeasible problem in at least one instance (state box / terminal equality)")
    if strict && !ALLOW_UNSOLVED[] && any(==(1), status)
        error("calculate!: $(count(==(1), status)) instance(s) without an optimality certificate (set AlmpcHIP.ALLOW_UNSOLVED[] = true to read the iterate)")
    end
    return status
end

# calculate!(C): fills u, e_u (m x N x batch) and x, e_x (n x (N+1) x batch) in place (src/main/computation_mpc.jl:38-55).
# first_move_only = true: only u[:, 1, :] is brought back (m x batch values, written into the first stage of `u`; 131 KB instead of
# 32 MB at the benchmark shape) -- what a receding-horizon caller applies; x, e_x, e_u are left untouched.
function calculate!(mod::HipModeler, x::Array{Float64}, e_x::Array{Float64}, u::Array{Float64}, e_u::Array{Float64};
                    first_move_only::Bool = false)
    check_result_sizes(mod, x, e_x, u, e_u)
    o = Ref(mod.opts)
    status = Vector{Int32}(undef, mod.batch)
    if mod.mode === :relin || mod.mode === :sqp
        if mod.mode === :relin
            relin_step!(mod)
        else
            length(mod.x0) == mod.n * mod.batch || error("calculate!: update_initialization! first")
            sqp_start!(mod, mod.x0)
            sqp_iterate!(mod, mod.sqp_iterations)
        end
        return throw_on_status(read_results!(mod, x, e_x, u, e_u); strict = mod.opts.polish != 0)
    end
    if first_move_only
        check(mod.handle, ccall((:almpc_calculate_async, libalmpc), Cint, (Ptr{Cvoid}, Ref{AlmpcOpts}), mod.handle, o))
        t = results_async(mod; u0 = true, status = true)
        u0 = Matrix{Float64}(undef, mod.m, mod.batch)
        results_wait!(mod, t; u0 = u0, status = status)
        ur = reshape(u, mod.m, mod.N, mod.batch)
        @views ur[:, 1, :] .= u0
    else
        check(mod.handle, ccall((:almpc_calculate, libalmpc), Cint, (Ptr{Cvoid}, Ref{AlmpcOpts}), mod.handle, o))
        check(mod.handle, ccall((:almpc_get_results, libalmpc), Cint,
                                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                                mod.handle, x, e_x, u, e_u, status, C_NULL, C_NULL))
    end
    return throw_on_status(status)
end

# ---- host-facing step path: pinned staging owned by the handle, transfers on copy streams (include/almpc.h) ----
"x0 (n x batch) -> pinned slot -> upload on the copy-in stream; returns at once, the next step waits for it on the device"
function update_initialization_async!(mod::HipModeler, x0::VecOrMat{Float64})
    length(x0) == mod.n * mod.batch || throw(DimensionMismatch("x0 must hold n x batch values"))
    check(mod.handle, ccall((:almpc_update_initialization_async, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float64}), mod.handle, x0))
end

"zero-copy input: an n x batch matrix over the pinned slot the next `update_initialization_async!` will use -- write the states into it (e.g.
straight from the plant simulation) and pass it on: the staging copy is skipped.  Valid until that call."
function x0_staging(mod::HipModeler)
    slot = Ref{Ptr{Float64}}(C_NULL)
    check(mod.handle, ccall((:almpc_x0_staging, libalmpc), Cint, (Ptr{Cvoid}, Ref{Ptr{Float64}}), mod.handle, slot))
    return unsafe_wrap(Array, slot[], (mod.n, mod.batch))
end

const WANT_X, WANT_E_X, WANT_U, WANT_E_U, WANT_STATUS, WANT_ITERS, WANT_POLISH_ITERS, WANT_FIRST_INPUT =
    UInt32(0x01), UInt32(0x02), UInt32(0x04), UInt32(0x08), UInt32(0x10), UInt32(0x20), UInt32(0x40), UInt32(0x80)

"ask for results of the last enqueued step (read-back on the copy-out stream, the next step may start meanwhile); returns a ticket"
function results_async(mod::HipModeler; x::Bool = false, e_x::Bool = false, u::Bool = false, e_u::Bool = false, u0::Bool = false,
                       status::Bool = false, iters::Bool = false, polish_iters::Bool = false)
    want = (x ? WANT_X : UInt32(0)) | (e_x ? WANT_E_X : UInt32(0)) | (u ? WANT_U : UInt32(0)) | (e_u ? WANT_E_U : UInt32(0)) |
           (u0 ? WANT_FIRST_INPUT : UInt32(0)) | (status ? WANT_STATUS : UInt32(0)) | (iters ? WANT_ITERS : UInt32(0)) |
           (polish_iters ? WANT_POLISH_ITERS : UInt32(0))
    t = ccall((:almpc_get_results_async, libalmpc), Cint, (Ptr{Cvoid}, UInt32), mod.handle, want)
    t < 0 && check(mod.handle, t)
    return t
end

"wait for a ticket and copy the given arrays out of the pinned slot (`nothing`: not wanted); sizes are checked"
function results_wait!(mod::HipModeler, ticket::Integer; x = nothing, e_x = nothing, u = nothing, e_u = nothing, u0 = nothing,
                       status = nothing, iters = nothing, polish_iters = nothing)
    check_result_sizes(mod, x, e_x, u, e_u)
    u0 === nothing || length(u0) == mod.m * mod.batch || throw(DimensionMismatch("u0 must hold m x batch values"))
    for (name, a) in (("status", status), ("iters", iters), ("polish_iters", polish_iters))
        a === nothing || length(a) == mod.batch || throw(DimensionMismatch("$name must hold batch values"))
    end
    pf(a) = a === nothing ? Ptr{Float64}(C_NULL) : pointer(a)
    pi(a) = a === nothing ? Ptr{Int32}(C_NULL) : pointer(a)
    GC.@preserve x e_x u e_u u0 status iters polish_iters check(mod.handle, ccall((:almpc_get_results_wait, libalmpc), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
        mod.handle, ticket, pf(x), pf(e_x), pf(u), pf(e_u), pf(u0), pi(status), pi(iters), pi(polish_iters)))
end

"zero-copy views of a ticket's pinned slot (after `results_wait!(mod, ticket)`): arrays that were not asked for come back as `nothing`; valid until two more requests"
function host_results(mod::HipModeler, ticket::Integer)
    pd = [Ref{Ptr{Float64}}(C_NULL) for _ in 1:5]; pn = [Ref{Ptr{Int32}}(C_NULL) for _ in 1:3]
    check(mod.handle, ccall((:almpc_host_results, libalmpc), Cint,
        (Ptr{Cvoid}, Cint, Ref{Ptr{Float64}}, Ref{Ptr{Float64}}, Ref{Ptr{Float64}}, Ref{Ptr{Float64}}, Ref{Ptr{Float64}}, Ref{Ptr{Int32}},
         Ref{Ptr{Int32}}, Ref{Ptr{Int32}}), mod.handle, ticket, pd[1], pd[2], pd[3], pd[4], pd[5], pn[1], pn[2], pn[3]))
    w(p, dims) = p[] == C_NULL ? nothing : unsafe_wrap(Array, p[], dims)
    return (x = w(pd[1], (mod.n, mod.N + 1, mod.batch)), e_x = w(pd[2], (mod.n, mod.N + 1, mod.batch)), u = w(pd[3], (mod.m, mod.N, mod.batch)),
            e_u = w(pd[4], (mod.m, mod.N, mod.batch)), u0 = w(pd[5], (mod.m, mod.batch)), status = w(pn[1], (mod.batch,)),
            iters = w(pn[2], (mod.batch,)), polish_iters = w(pn[3], (mod.batch,)))
end

"u[:, 1] of every instance of the last step, m x batch (`almpc_get_first_input`)"
function first_input!(mod::HipModeler, u0::Matrix{Float64})
    length(u0) == mod.m * mod.batch || throw(DimensionMismatch("u0 must hold m x batch values"))
    check(mod.handle, ccall((:almpc_get_first_input, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float64}), mod.handle, u0))
    return u0
end
first_input(mod::HipModeler) = first_input!(mod, Matrix{Float64}(undef, mod.m, mod.batch))

"""
    design_relin_fnn!(mod, W_in, W_h, b_h, W_out, activation, Q, R, S, P, umin, umax; x_ref, u_ref)

BASELINE configs[3]: the black-box model is re-linearised at every instance's own state in every step, on the device
(`almpc_relin_fnn_*`).  `P` as the reference takes it: DARE at the linearisation about the LAST reference
(src/sub/design_mpc.jl:312-327).  Then per step: `update_initialization!(mod, X0)`; `relin_step!(mod)`; `read_results!`.
"""
function design_relin_fnn!(mod::HipModeler, W_in::Matrix{Float64}, W_h::Array{Float64,3}, b_h::Matrix{Float64}, W_out::Matrix{Float64},
                           activation::Integer, Q::Matrix{Float64}, R::Matrix{Float64}, S::Matrix{Float64}, P::Matrix{Float64},
                           umin::Vector{Float64}, umax::Vector{Float64}; x_ref::Matrix{Float64}, u_ref::Matrix{Float64})
    check(mod.handle, ccall((:almpc_relin_fnn_setup, libalmpc), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble),
                   mod.handle, size(W_in, 1), size(W_h, 3), activation, W_in, W_h, b_h, W_out, x_ref, u_ref, Q, R, S, P, umin, umax,
                   mod.opts.rho, mod.opts.sigma))
    return mod
end

"one step; `warm = true` after a solved step: working-set guess from the previous inputs shifted one stage, no ADMM phase"
function relin_step!(mod::HipModeler; warm::Bool = false)
    o0 = mod.opts
    o = Ref(AlmpcOpts(o0.rho, o0.sigma, o0.alpha, o0.eps_abs, o0.eps_rel, o0.max_iter, o0.check_every, o0.polish, o0.polish_max_iter,
                      Int32(warm), o0.reserved))
    check(mod.handle, ccall((:almpc_relin_fnn_step, libalmpc), Cint, (Ptr{Cvoid}, Ref{AlmpcOpts}), mod.handle, o))
end

"x0 <- fnn(x0, u[:,1]) on the device: the closed loop of the black-box model itself (simulation studies)"
relin_advance!(mod::HipModeler) = check(mod.handle, ccall((:almpc_relin_fnn_advance, libalmpc), Cint, (Ptr{Cvoid},), mod.handle))

"copy the results of the last step into caller-owned arrays (m x N x batch, n x (N+1) x batch); returns the per-instance status"
function read_results!(mod::HipModeler, x::Array{Float64}, e_x::Array{Float64}, u::Array{Float64}, e_u::Array{Float64})
    check_result_sizes(mod, x, e_x, u, e_u)
    status = Vector{Int32}(undef, mod.batch)
    check(mod.handle, ccall((:almpc_get_results, libalmpc), Cint,
                            (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                            mod.handle, x, e_x, u, e_u, status, C_NULL, C_NULL))
    return status
end

# ---- multi-GPU: one Julia process per GPU, each with its own handle on its shard of the batch; RCCL inside the library ----
"rank 0 makes the id and hands the 128 bytes to the other ranks (MPI.jl, Distributed, a file: any channel)"
function comm_unique_id()
    id = Vector{UInt8}(undef, 128)
    rc = ccall((:almpc_comm_unique_id, libalmpc), Cint, (Ptr{UInt8},), id)
    rc == 0 || error("almpc_comm_unique_id failed ($rc): librccl not loadable?")
    return id
end

comm_init!(mod::HipModeler, id::Vector{UInt8}, rank::Integer, world::Integer) =
    check(mod.handle, ccall((:almpc_comm_init, libalmpc), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), mod.handle, id, rank, world))

"(ranks, unsolved instances over all ranks, max ADMM iterations, max polish iterations) of the last step, all-reduced over RCCL"
function comm_summary(mod::HipModeler)
    out = Vector{Int64}(undef, 4)
    check(mod.handle, ccall((:almpc_comm_summary, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Int64}), mod.handle, out))
    return out
end

"u[:, 1] of every instance of every rank (m x batch x world), all-gathered over RCCL: what a plant simulator on any rank needs next"
function comm_allgather_first_input(mod::HipModeler, world::Integer)
    out = Array{Float64,3}(undef, mod.m, mod.batch, world)
    check(mod.handle, ccall((:almpc_comm_allgather_first_input, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Ptr{Float64}}),
                            mod.handle, out, C_NULL))
    return out
end

# ---- the rest of the ABI: asynchronous stepping, timing, parity hooks, device-resident results, helpers ----
"launch a step on the handle's stream and return at once; `synchronize!` waits for it"
function calculate_async!(mod::HipModeler)
    o = Ref(mod.opts)
    check(mod.handle, ccall((:almpc_calculate_async, libalmpc), Cint, (Ptr{Cvoid}, Ref{AlmpcOpts}), mod.handle, o))
end
synchronize!(mod::HipModeler) = check(mod.handle, ccall((:almpc_synchronize, libalmpc), Cint, (Ptr{Cvoid},), mod.handle))
function relin_step_async!(mod::HipModeler)
    o = Ref(mod.opts)
    check(mod.handle, ccall((:almpc_relin_fnn_step_async, libalmpc), Cint, (Ptr{Cvoid}, Ref{AlmpcOpts}), mod.handle, o))
end
"x0 <- A x0 + B u[:,1] on the device (shared-model designs): the closed loop without a host round trip"
advance_plant!(mod::HipModeler) = check(mod.handle, ccall((:almpc_advance_plant, libalmpc), Cint, (Ptr{Cvoid},), mod.handle))
"x0 already in device memory (`d_x0`: device pointer to batch x n doubles)"
update_initialization_device!(mod::HipModeler, d_x0::Ptr{Float64}) =
    check(mod.handle, ccall((:almpc_update_initialization_device, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float64}), mod.handle, d_x0))
"device pointers of the result arrays of the last step (x, e_x, u, e_u): for callers that keep the loop on the GPU"
function device_results(mod::HipModeler)
    px, pex, pu, peu = Ref{Ptr{Float64}}(C_NULL), Ref{Ptr{Float64}}(C_NULL), Ref{Ptr{Float64}}(C_NULL), Ref{Ptr{Float64}}(C_NULL)
    check(mod.handle, ccall((:almpc_device_results, libalmpc), Cint,
                            (Ptr{Cvoid}, Ref{Ptr{Float64}}, Ref{Ptr{Float64}}, Ref{Ptr{Float64}}, Ref{Ptr{Float64}}), mod.handle, px, pex, pu, peu))
    return px[], pex[], pu[], peu[]
end
"structured handle: the next step starts from `src`'s last inputs (same batch, horizon <= this one's): horizon continuation"
start_from!(mod::HipModeler, src::HipModeler) =
    check(mod.handle, ccall((:almpc_set_start_from, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), mod.handle, src.handle))
set_step_fusion!(mod::HipModeler, on::Bool) = check(mod.handle, ccall((:almpc_set_step_fusion, libalmpc), Cint, (Ptr{Cvoid}, Cint), mod.handle, on ? 1 : 0))
"instances the condensed path leaves unsolved (open-loop unstable linearisations) are redone in the multiple-shooting form; before the design"
set_structured_fallback!(mod::HipModeler, on::Bool) =
    check(mod.handle, ccall((:almpc_set_structured_fallback, libalmpc), Cint, (Ptr{Cvoid}, Cint), mod.handle, on ? 1 : 0))
"instances whose SQP iteration was skipped (indefinite condensed Hessian, non-finite or infeasible QP)"
function sqp_skipped(mod::HipModeler)
    out = Vector{Int32}(undef, mod.batch)
    check(mod.handle, ccall((:almpc_sqp_fnn_skipped, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Int32}), mod.handle, out))
    return out
end
"(H_i, F_i, d_i) of one instance of a per-instance design; `gradient_instance`: q_i of a time-varying design (parity hooks)"
function design_instance(mod::HipModeler, i::Integer)
    nz = mod.m * mod.N
    H, F, d = Matrix{Float64}(undef, nz, nz), Matrix{Float64}(undef, nz, mod.n), Vector{Float64}(undef, nz)
    check(mod.handle, ccall((:almpc_get_design_instance, libalmpc), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                            mod.handle, i - 1, H, F, d))
    return H, F, d
end
function gradient_instance(mod::HipModeler, i::Integer)
    q = Vector{Float64}(undef, mod.m * mod.N)
    check(mod.handle, ccall((:almpc_get_gradient_instance, libalmpc), Cint, (Ptr{Cvoid}, Cint, Ptr{Float64}), mod.handle, i - 1, q))
    return q
end
"""
    design_ltv!(mod, A_all, B_all, c_all, xbar, ubar, Q, R, S, P, umin, umax; x_ref, u_ref)

Time-varying models per instance (`A_all` n x n x N x batch, `B_all` n x m x N x batch, `c_all` n x N x batch or `nothing`,
`xbar` n x (N+1) x batch, `ubar` m x N x batch): the QP of one SQP / multiple-shooting iteration in v = u - ubar (`almpc_design_ltv`).
"""
function design_ltv!(mod::HipModeler, A_all::Array{Float64,4}, B_all::Array{Float64,4}, c_all, xbar::Array{Float64,3}, ubar::Array{Float64,3},
                     Q::Matrix{Float64}, R::Matrix{Float64}, S::Matrix{Float64}, P::Matrix{Float64}, umin::Vector{Float64}, umax::Vector{Float64};
                     x_ref::Matrix{Float64}, u_ref::Matrix{Float64})
    pc = c_all === nothing ? Ptr{Float64}(C_NULL) : pointer(c_all)
    GC.@preserve c_all check(mod.handle, ccall((:almpc_design_ltv, libalmpc), Cint,
                   (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble),
                   mod.handle, A_all, B_all, pc, xbar, ubar, x_ref, u_ref, Q, R, S, P, 0, umin, umax, mod.opts.rho, mod.opts.sigma))
    return mod
end
"Jacobians (A_i, B_i) and values of an Fnn at `x` (n x batch), `u` (m x batch) on device `device` (the batched `proceed_system_linearization`)"
function fnn_linearize(W_in::Matrix{Float64}, W_h::Array{Float64,3}, b_h::Matrix{Float64}, W_out::Matrix{Float64}, activation::Integer,
                       x::Matrix{Float64}, u::Matrix{Float64}; device::Integer = 0)
    n, m, b = size(x, 1), size(u, 1), size(x, 2)
    A, B, f = Array{Float64,3}(undef, n, n, b), Array{Float64,3}(undef, n, m, b), Matrix{Float64}(undef, n, b)
    rc = ccall((:almpc_fnn_linearize, libalmpc), Cint,
               (Cint, Cint, Cint, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64},
                Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
               device, n, m, size(W_in, 1), size(W_h, 3), activation, W_in, W_h, b_h, W_out, b, x, u, A, B, f)
    rc == 0 || error("almpc_fnn_linearize failed ($rc)")
    return A, B, f
end
"P = DARE(A, B, Q, R) as the library computes it (host; src/sub/design_mpc.jl:327)"
function dare(A::Matrix{Float64}, B::Matrix{Float64}, Q::Matrix{Float64}, R::Matrix{Float64})
    n, m = size(B)
    P = Matrix{Float64}(undef, n, n)
    rc = ccall((:almpc_dare, libalmpc), Cint, (Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), n, m, A, B, Q, R, P)
    rc == 0 || error("almpc_dare: no convergence ($rc)")
    return P
end
"library defaults of the options (OSQP's, plus the documented changes)"
function default_opts()
    o = Ref(AlmpcOpts())
    ccall((:almpc_default_opts, libalmpc), Cvoid, (Ref{AlmpcOpts},), o)
    return o[]
end
# timing (handles created with ALMPC_FLAG_TIMING = 0x1): per-stage milliseconds of the last step / sums since the last reset
function get_timing(mod::HipModeler)
    t = zeros(Float32, 4)
    check(mod.handle, ccall((:almpc_get_timing, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                            mod.handle, pointer(t, 1), pointer(t, 2), pointer(t, 3), pointer(t, 4)))
    return (admm_ms = t[1], polish_ms = t[2], rollout_ms = t[3], total_ms = t[4])
end
timing_reset!(mod::HipModeler, reserve_steps::Integer = 0) =
    check(mod.handle, ccall((:almpc_timing_reset, libalmpc), Cint, (Ptr{Cvoid}, Cint), mod.handle, reserve_steps))
timing_set_stride!(mod::HipModeler, every::Integer) =
    check(mod.handle, ccall((:almpc_timing_set_stride, libalmpc), Cint, (Ptr{Cvoid}, Cint), mod.handle, every))
function timing_summary(mod::HipModeler)
    steps = Ref{Cint}(0); t = zeros(Float64, 4)
    check(mod.handle, ccall((:almpc_timing_summary, libalmpc), Cint, (Ptr{Cvoid}, Ref{Cint}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                            mod.handle, steps, pointer(t, 1), pointer(t, 2), pointer(t, 3), pointer(t, 4)))
    return (steps = Int(steps[]), admm_ms = t[1], polish_ms = t[2], rollout_ms = t[3], total_ms = t[4])
end
"per recorded step its four stage times in ms (rows: admm, polish, rollout, total), at most `cap` steps"
function timing_samples(mod::HipModeler, cap::Integer = 4096)
    cnt = Ref{Cint}(0); t = zeros(Float32, cap, 4)
    check(mod.handle, ccall((:almpc_timing_samples, libalmpc), Cint, (Ptr{Cvoid}, Cint, Ref{Cint}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                            mod.handle, cap, cnt, pointer(t, 1), pointer(t, cap + 1), pointer(t, 2cap + 1), pointer(t, 3cap + 1)))
    return t[1:min(cap, Int(cnt[])), :]
end
function relin_timing(mod::HipModeler)
    t = zeros(Float32, 3)
    check(mod.handle, ccall((:almpc_relin_fnn_timing, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                            mod.handle, pointer(t, 1), pointer(t, 2), pointer(t, 3)))
    return (jacobian_ms = t[1], design_ms = t[2], step_ms = t[3])
end
"diagnostic: fill the LDS of every CU with NaN patterns before a step (a kernel that reads LDS it did not write then shows)"
debug_poison_lds!(mod::HipModeler) = check(mod.handle, ccall((:almpc_debug_poison_lds, libalmpc), Cint, (Ptr{Cvoid},), mod.handle))

# ---- one Julia process, several GPUs (almpc_group_*): the reference API is one process, one call (src/main/main_mpc.jl:22-53) ----
mutable struct HipGroup
    group::Ptr{Cvoid}
    n::Int; m::Int; N::Int; batch::Int
    opts::AlmpcOpts
end

function gcheck(g, rc)
    rc == 0 && return
    error("libalmpc group error $rc: " * unsafe_string(ccall((:almpc_group_last_error, libalmpc), Cstring, (Ptr{Cvoid},), g)))
end

"handle i (1-based) of the group as a HipModeler on its shard: for the per-handle options and anything the group calls do not cover"
function group_handle(g::HipGroup, i::Integer)
    h = ccall((:almpc_group_handle, libalmpc), Ptr{Cvoid}, (Ptr{Cvoid}, Cint), g.group, i - 1)
    h == C_NULL && throw(BoundsError(g, i))
    return HipModeler(h, g.n, g.m, g.N, group_shard(g, i)[2], g.opts)    # (no finalizer: the group owns the handle)
end
"(first instance (1-based), count) of handle i's contiguous shard"
function group_shard(g::HipGroup, i::Integer)
    f, c = Ref{Cint}(0), Ref{Cint}(0)
    gcheck(g.group, ccall((:almpc_group_shard, libalmpc), Cint, (Ptr{Cvoid}, Cint, Ref{Cint}, Ref{Cint}), g.group, i - 1, f, c))
    return Int(f[]) + 1, Int(c[])
end

"`design_hip` over several devices: `devices` lists one HIP device id per shard (`mpc_batch` instances are cut into contiguous shards)"
function group_design_hip(A::Matrix{Float64}, B::Matrix{Float64}, Q::Matrix{Float64}, R::Matrix{Float64}, S::Matrix{Float64},
                          umin::Vector{Float64}, umax::Vector{Float64}, N::Int; batch::Int, devices::Vector{Int},
                          x_ref::Matrix{Float64}, u_ref::Matrix{Float64}, opts::AlmpcOpts = AlmpcOpts(),
                          xmin::Union{Nothing,Vector{Float64}} = nothing, xmax::Union{Nothing,Vector{Float64}} = nothing,
                          terminal::String = "none", rho_profile::String = "scalar", P::Union{Nothing,Matrix{Float64}} = nothing)
    n, m = size(B)
    terminal == "contractive" && error("terminal ingredient \"contractive\" is a quadratic constraint (src/sub/design_mpc.jl:333-340), not a QP row")
    gref = Ref{Ptr{Cvoid}}(C_NULL)
    devs = Cint.(devices)
    rc = ccall((:almpc_group_create, libalmpc), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Cint, Cint, Cint, Ptr{Cint}, UInt32),
               gref, n, m, N, batch, length(devs), devs, 0)
    rc == 0 || error("almpc_group_create failed ($rc): are the devices visible? (there is no CPU fallback)")
    g = HipGroup(gref[], n, m, N, batch, opts)
    finalizer(x -> ccall((:almpc_group_destroy, libalmpc), Cvoid, (Ptr{Cvoid},), x.group), g)
    for i in 1:ccall((:almpc_group_size, libalmpc), Cint, (Ptr{Cvoid},), g.group)
        h = group_handle(g, i).handle
        check(h, ccall((:almpc_set_terminal_equality, libalmpc), Cint, (Ptr{Cvoid}, Cint), h, terminal == "equality" ? 1 : 0))
        check(h, ccall((:almpc_set_rho_profile, libalmpc), Cint, (Ptr{Cvoid}, Cint), h, rho_profile == "stiffness" ? 1 : 0))
    end
    pP = P === nothing ? Ptr{Float64}(C_NULL) : pointer(P)
    pxmin = xmin === nothing ? Ptr{Float64}(C_NULL) : pointer(xmin)
    pxmax = xmax === nothing ? Ptr{Float64}(C_NULL) : pointer(xmax)
    GC.@preserve P xmin xmax gcheck(g.group, ccall((:almpc_group_design_shared, libalmpc), Cint,
                   (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble),
                   g.group, A, B, Q, R, S, pP, umin, umax, pxmin, pxmax, opts.rho, opts.sigma))
    gcheck(g.group, ccall((:almpc_group_set_reference, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cint), g.group, x_ref, u_ref, 0))
    return g
end

function group_update_initialization!(g::HipGroup, x0::VecOrMat{Float64})
    length(x0) == g.n * g.batch || throw(DimensionMismatch("x0 must hold n x batch values"))
    gcheck(g.group, ccall((:almpc_group_update_initialization, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float64}), g.group, x0))
end
function group_calculate_async!(g::HipGroup)
    o = Ref(g.opts)
    gcheck(g.group, ccall((:almpc_group_calculate_async, libalmpc), Cint, (Ptr{Cvoid}, Ref{AlmpcOpts}), g.group, o))
end
group_synchronize!(g::HipGroup) = gcheck(g.group, ccall((:almpc_group_synchronize, libalmpc), Cint, (Ptr{Cvoid},), g.group))
function group_calculate!(g::HipGroup)
    o = Ref(g.opts)
    gcheck(g.group, ccall((:almpc_group_calculate, libalmpc), Cint, (Ptr{Cvoid}, Ref{AlmpcOpts}), g.group, o))
end
"results of the whole batch (any array may be `nothing`); `u0`: m x batch first inputs; returns the per-instance status"
function group_read_results!(g::HipGroup; x = nothing, e_x = nothing, u = nothing, e_u = nothing, u0 = nothing)
    nx, nu = g.n * (g.N + 1) * g.batch, g.m * g.N * g.batch
    for (name, a, want) in (("x", x, nx), ("e_x", e_x, nx), ("u", u, nu), ("e_u", e_u, nu), ("u0", u0, g.m * g.batch))
        a === nothing || length(a) == want || throw(DimensionMismatch("$name holds $(length(a)) values, the group writes $want"))
    end
    status = Vector{Int32}(undef, g.batch)
    pf(a) = a === nothing ? Ptr{Float64}(C_NULL) : pointer(a)
    GC.@preserve x e_x u e_u u0 gcheck(g.group, ccall((:almpc_group_get_results, libalmpc), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
        g.group, pf(x), pf(e_x), pf(u), pf(e_u), pf(u0), status, C_NULL, C_NULL))
    return throw_on_status(status; strict = g.opts.polish != 0)
end


# ---- group forms of everything a handle can do (include/almpc.h: almpc_group_*) ----
"options of every handle of the group (take effect at the next design)"
function group_set_state_rows!(g::HipGroup; xmin::Union{Nothing,Vector{Float64}} = nothing, xmax::Union{Nothing,Vector{Float64}} = nothing,
                               terminal::String = "none")
    (xmin === nothing) == (xmax === nothing) || error("give both xmin and xmax or neither")
    gcheck(g.group, ccall((:almpc_group_set_terminal_equality, libalmpc), Cint, (Ptr{Cvoid}, Cint), g.group, terminal == "equality" ? 1 : 0))
    pmin = xmin === nothing ? Ptr{Float64}(C_NULL) : pointer(xmin)
    pmax = xmax === nothing ? Ptr{Float64}(C_NULL) : pointer(xmax)
    GC.@preserve xmin xmax gcheck(g.group, ccall((:almpc_group_set_state_box, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), g.group, pmin, pmax))
    return g
end
group_set_rho_profile!(g::HipGroup, profile::String) =
    gcheck(g.group, ccall((:almpc_group_set_rho_profile, libalmpc), Cint, (Ptr{Cvoid}, Cint), g.group, profile == "stiffness" ? 1 : 0))
group_set_structured_fallback!(g::HipGroup, on::Bool) =
    gcheck(g.group, ccall((:almpc_group_set_structured_fallback, libalmpc), Cint, (Ptr{Cvoid}, Cint), g.group, on ? 1 : 0))

"`design_batched!` for the whole batch: `A_batch` n x n x batch, `B_batch` n x m x batch, `P` nothing | n x n | n x n x batch"
function group_design_batched!(g::HipGroup, A_batch::Array{Float64,3}, B_batch::Array{Float64,3}, Q::Matrix{Float64}, R::Matrix{Float64},
                               S::Matrix{Float64}, P, umin::Vector{Float64}, umax::Vector{Float64}; x_ref::Matrix{Float64}, u_ref::Matrix{Float64})
    size(A_batch, 3) == g.batch && size(B_batch, 3) == g.batch || throw(DimensionMismatch("one model per instance"))
    pptr = P === nothing ? Ptr{Float64}(C_NULL) : pointer(P)
    pinst = (P !== nothing && ndims(P) == 3) ? 1 : 0
    GC.@preserve P gcheck(g.group, ccall((:almpc_group_design_batched, libalmpc), Cint,
                   (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint,
                    Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble),
                   g.group, A_batch, B_batch, Q, R, S, pptr, pinst, umin, umax, g.opts.rho, g.opts.sigma))
    gcheck(g.group, ccall((:almpc_group_set_reference, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cint), g.group, x_ref, u_ref, 0))
    return g
end

"`design_relin_fnn!` on every device (BASELINE configs[3]); then `group_relin_step!`, `group_relin_advance!`"
function group_design_relin_fnn!(g::HipGroup, W_in::Matrix{Float64}, W_h::Array{Float64,3}, b_h::Matrix{Float64}, W_out::Matrix{Float64},
                                 activation::Integer, Q::Matrix{Float64}, R::Matrix{Float64}, S::Matrix{Float64}, P::Matrix{Float64},
                                 umin::Vector{Float64}, umax::Vector{Float64}; x_ref::Matrix{Float64}, u_ref::Matrix{Float64})
    gcheck(g.group, ccall((:almpc_group_relin_fnn_setup, libalmpc), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble),
                   g.group, size(W_in, 1), size(W_h, 3), activation, W_in, W_h, b_h, W_out, x_ref, u_ref, Q, R, S, P, umin, umax,
                   g.opts.rho, g.opts.sigma))
    return g
end
function group_relin_step!(g::HipGroup; warm::Bool = false, async::Bool = false)
    o0 = g.opts
    o = Ref(AlmpcOpts(o0.rho, o0.sigma, o0.alpha, o0.eps_abs, o0.eps_rel, o0.max_iter, o0.check_every, o0.polish, o0.polish_max_iter,
                      Int32(warm), o0.reserved))
    if async
        gcheck(g.group, ccall((:almpc_group_relin_fnn_step_async, libalmpc), Cint, (Ptr{Cvoid}, Ref{AlmpcOpts}), g.group, o))
    else
        gcheck(g.group, ccall((:almpc_group_relin_fnn_step, libalmpc), Cint, (Ptr{Cvoid}, Ref{AlmpcOpts}), g.group, o))
    end
end
group_relin_advance!(g::HipGroup) = gcheck(g.group, ccall((:almpc_group_relin_fnn_advance, libalmpc), Cint, (Ptr{Cvoid},), g.group))
group_advance_plant!(g::HipGroup) = gcheck(g.group, ccall((:almpc_group_advance_plant, libalmpc), Cint, (Ptr{Cvoid},), g.group))

"`design_sqp_fnn!` on every device (BASELINE configs[4]); `P` n x n or n x n x batch"
function group_design_sqp_fnn!(g::HipGroup, W_in::Matrix{Float64}, W_h::Array{Float64,3}, b_h::Matrix{Float64}, W_out::Matrix{Float64},
                               activation::Integer, Q::Matrix{Float64}, R::Matrix{Float64}, S::Matrix{Float64}, P::Array{Float64},
                               umin::Vector{Float64}, umax::Vector{Float64}; x_ref::Matrix{Float64}, u_ref::Matrix{Float64},
                               structured_qp::Bool = false)
    gcheck(g.group, ccall((:almpc_group_sqp_fnn_set_structured, libalmpc), Cint, (Ptr{Cvoid}, Cint), g.group, structured_qp ? 1 : 0))
    gcheck(g.group, ccall((:almpc_group_sqp_fnn_setup, libalmpc), Cint,
                   (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble),
                   g.group, size(W_in, 1), size(W_h, 3), activation, W_in, W_h, b_h, W_out, x_ref, u_ref, Q, R, S, P, ndims(P) == 3 ? 1 : 0,
                   umin, umax, g.opts.rho, g.opts.sigma))
    return g
end
function group_sqp_start!(g::HipGroup, x0::Matrix{Float64}; u_guess::Union{Nothing,Array{Float64,3}} = nothing)
    length(x0) == g.n * g.batch || throw(DimensionMismatch("x0 must hold n x batch values"))
    pg = u_guess === nothing ? Ptr{Float64}(C_NULL) : pointer(u_guess)
    GC.@preserve u_guess gcheck(g.group, ccall((:almpc_group_sqp_fnn_start, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), g.group, x0, pg))
end
function group_sqp_iterate!(g::HipGroup, iters::Integer; step::Float64 = 1.0, merit_safeguard::Bool = true)
    gcheck(g.group, ccall((:almpc_group_sqp_fnn_set_step_rule, libalmpc), Cint, (Ptr{Cvoid}, Cint), g.group, merit_safeguard ? 1 : 0))
    st, de = zeros(iters), zeros(iters)
    o = Ref(g.opts)
    gcheck(g.group, ccall((:almpc_group_sqp_fnn_iterate, libalmpc), Cint, (Ptr{Cvoid}, Cint, Cdouble, Ref{AlmpcOpts}, Ptr{Float64}, Ptr{Float64}),
                          g.group, iters, step, o, st, de))
    return st, de
end
function group_sqp_skipped(g::HipGroup)
    sk = Vector{Int32}(undef, g.batch)
    gcheck(g.group, ccall((:almpc_group_sqp_fnn_skipped, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Int32}), g.group, sk))
    return sk
end

"zero-copy input: the handles' pinned x0 slots (n x count_i matrices over library memory); write the states, then `group_update_initialization_staged!`"
function group_x0_staging(g::HipGroup)
    k = ccall((:almpc_group_size, libalmpc), Cint, (Ptr{Cvoid},), g.group)
    slots = Vector{Ptr{Float64}}(undef, k)
    gcheck(g.group, ccall((:almpc_group_x0_staging, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Ptr{Float64}}), g.group, slots))
    return slots, [unsafe_wrap(Array, slots[i], (g.n, group_shard(g, i)[2])) for i in 1:k]
end
group_update_initialization_staged!(g::HipGroup, slots::Vector{Ptr{Float64}}) =
    gcheck(g.group, ccall((:almpc_group_update_initialization_staged, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Ptr{Float64}}), g.group, slots))

"request the results of the last enqueued step of every device (`want`: mask of ALMPC_WANT_*); returns the group's ticket"
function group_results_async(g::HipGroup, want::Integer)
    # `group_results_wait!` always reads the status (solution or throw, src/main/computation_mpc.jl:41-53): it is part of every request
    t = ccall((:almpc_group_get_results_async, libalmpc), Cint, (Ptr{Cvoid}, UInt32), g.group, UInt32(want) | WANT_STATUS)
    t < 0 && gcheck(g.group, t)
    return t
end
"wait for a ticket of `group_results_async` and gather into the caller's arrays (any may be `nothing`); returns the status vector"
function group_results_wait!(g::HipGroup, ticket::Integer; x = nothing, e_x = nothing, u = nothing, e_u = nothing, u0 = nothing)
    nx, nu = g.n * (g.N + 1) * g.batch, g.m * g.N * g.batch
    for (name, a, want) in (("x", x, nx), ("e_x", e_x, nx), ("u", u, nu), ("e_u", e_u, nu), ("u0", u0, g.m * g.batch))
        a === nothing || length(a) == want || throw(DimensionMismatch("$name holds $(length(a)) values, the group writes $want"))
    end
    status = Vector{Int32}(undef, g.batch)
    pf(a) = a === nothing ? Ptr{Float64}(C_NULL) : pointer(a)
    GC.@preserve x e_x u e_u u0 gcheck(g.group, ccall((:almpc_group_get_results_wait, libalmpc), Cint,
        (Ptr{Cvoid}, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
        g.group, ticket, pf(x), pf(e_x), pf(u), pf(e_u), pf(u0), status, C_NULL, C_NULL))
    return throw_on_status(status; strict = g.opts.polish != 0)
end

# the name BASELINE.json uses; absent from the reference (SURVEY.md section 0): one batched step, host in / host out
function _model_predictive_control_computation(mod::HipModeler, X0::Matrix{Float64}, x, e_x, u, e_u; first_move_only::Bool = false)
    check_result_sizes(mod, x, e_x, u, e_u)
    if mod.mode === :sqp
        update_initialization!(mod, X0)      # the SQP loop restarts from mod.x0 (almpc_sqp_fnn_start), which only this method stores
    else
        update_initialization_async!(mod, X0)
    end
    calculate!(mod, x, e_x, u, e_u; first_move_only = first_move_only)
end
function _model_predictive_control_computation(g::HipGroup, X0::Matrix{Float64}, x, e_x, u, e_u)
    group_update_initialization!(g, X0)
    group_calculate!(g)
    group_read_results!(g; x = x, e_x = e_x, u = u, e_u = e_u)
end

end # module
