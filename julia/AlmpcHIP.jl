# AlmpcHIP.jl -- thin Julia shim over libalmpc.so (include/almpc.h).
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no `julia`.  Every behaviour of the ABI is exercised
# through the same entry points from Python (tests/test_gpu_parity.py); this file is the binding a maintainer of
# AutomationLabsModelPredictiveControl.jl would add.  It keeps the reference's names and call shapes
# (src/main/main_mpc.jl:22-53, src/sub/design_mpc.jl:54-129, src/main/computation_mpc.jl:17-55) and adds the
# solver tag "hip" next to osqp/scip/ipopt/auto (src/sub/solver_selection.jl:9-14).
module AlmpcHIP

export hip_solver_def, HipModeler, design_hip, design_batched!, design_sqp_fnn!, sqp_start!, sqp_iterate!,
       update_initialization!, calculate!,
       _model_predictive_control_computation

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
end

function check(h, rc)
    rc == 0 && return
    msg = unsafe_string(ccall((:almpc_last_error, libalmpc), Cstring, (Ptr{Cvoid},), h))
    error("libalmpc error $rc: $msg")   # the reference throws from JuMP.value when no solution exists
end

"""
    design_hip(A, B, Q, R, S, umin, umax, N; batch = 1, device = 0, x_ref, u_ref, opts)

Design for `ConstrainedLinearControlDiscreteSystem` (replaces the JuMP model built at
src/sub/model_modeler_implementation/linear/mpc_modeler_implementation_linear.jl:20-103 and the objective of
src/sub/design_mpc.jl:405-468).  `P` is computed inside (DARE, src/sub/design_mpc.jl:327).
"""
function design_hip(A::Matrix{Float64}, B::Matrix{Float64}, Q::Matrix{Float64}, R::Matrix{Float64},
                    S::Matrix{Float64}, umin::Vector{Float64}, umax::Vector{Float64}, N::Int;
                    batch::Int = 1, device::Int = 0, x_ref::Matrix{Float64}, u_ref::Matrix{Float64},
                    opts::AlmpcOpts = AlmpcOpts())
    n, m = size(B)
    href = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:almpc_create, libalmpc), Cint, (Ref{Ptr{Cvoid}}, Cint, Cint, Cint, Cint, Cint, UInt32),
               href, n, m, N, batch, device, 0)
    rc == 0 || error("almpc_create failed ($rc): no gfx950 device? (there is no CPU fallback)")
    h = href[]
    check(h, ccall((:almpc_design_shared, libalmpc), Cint,
                   (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64},
                    Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble),
                   h, A, B, Q, R, S, C_NULL, umin, umax, C_NULL, C_NULL, opts.rho, opts.sigma))
    # references: n x (N+1) and m x N Julia matrices are already the ABI's [N+1][n] / [N][m] memory
    check(h, ccall((:almpc_set_reference, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cint),
                   h, x_ref, u_ref, 0))
    mod = HipModeler(h, n, m, N, batch, opts)
    finalizer(x -> ccall((:almpc_destroy, libalmpc), Cvoid, (Ptr{Cvoid},), x.handle), mod)
    return mod
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
                         umin::Vector{Float64}, umax::Vector{Float64}; x_ref::Matrix{Float64}, u_ref::Matrix{Float64})
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
    check(mod.handle, ccall((:almpc_update_initialization, libalmpc), Cint, (Ptr{Cvoid}, Ptr{Float64}), mod.handle, x0))
end

# calculate!(C): fills u, e_u (m x N x batch) and x, e_x (n x (N+1) x batch) in place (src/main/computation_mpc.jl:38-55)
function calculate!(mod::HipModeler, x::Array{Float64}, e_x::Array{Float64}, u::Array{Float64}, e_u::Array{Float64})
    o = Ref(mod.opts)
    check(mod.handle, ccall((:almpc_calculate, libalmpc), Cint, (Ptr{Cvoid}, Ref{AlmpcOpts}), mod.handle, o))
    status = Vector{Int32}(undef, mod.batch)
    check(mod.handle, ccall((:almpc_get_results, libalmpc), Cint,
                            (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}),
                            mod.handle, x, e_x, u, e_u, status, C_NULL, C_NULL))
    any(==(2), status) && error("calculate!: non-finite values in at least one instance")
    return status
end

# the name BASELINE.json uses; absent from the reference (SURVEY.md section 0): one batched step
function _model_predictive_control_computation(mod::HipModeler, X0::Matrix{Float64}, x, e_x, u, e_u)
    update_initialization!(mod, X0)
    calculate!(mod, x, e_x, u, e_u)
end

end # module
