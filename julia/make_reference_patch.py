#!/usr/bin/env python3
"""Regenerate julia/reference_hip.patch: the reference-side edits that wire libalmpc.so into AutomationLabsModelPredictiveControl.jl.

    python julia/make_reference_patch.py [/root/reference]

Build container only (needs the reference checkout; nothing here runs on the GPU box).  The reference's own files are copied into a
scratch git repository, the edits below are inserted at anchor lines of those files, julia/AlmpcHIP.jl is added as src/hip/AlmpcHIP.jl,
and `git diff` of the result is the patch.  Only the ADDED lines are this repository's; context lines are the reference's by the nature
of a diff.  tests/test_host_logic.py checks that the committed patch applies to the reference and equals what this script produces.
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = ["src/types/types.jl", "src/sub/solver_selection.jl", "src/sub/design_mpc.jl", "src/main/computation_mpc.jl",
         "src/AutomationLabsModelPredictiveControl.jl"]

# (file, anchor text that must occur exactly once, where, inserted text)
EDITS = [
    ("src/AutomationLabsModelPredictiveControl.jl", 'include("types/types.jl")\n', "before",
     'include("hip/AlmpcHIP.jl")   # thin ccall shim over libalmpc.so (MI355X batched condensed-QP engine)\n'),

    ("src/types/types.jl", 'struct auto_solver_def <: AbstractSolvers end\n', "after",
     '\n"""\n    hip\nBatched condensed-QP solve on an AMD MI355X through libalmpc.so (src/hip/AlmpcHIP.jl): the linear method on a linear system, and\n'
     'black-box Fnn systems by linearisation (linear method) or Gauss-Newton SQP (non linear method).\n"""\n'
     'struct hip_solver_def <: AbstractSolvers end\n'),

    ("src/sub/solver_selection.jl", '    auto = auto_solver_def(),\n', "after", '    hip = hip_solver_def(),\n'),

    ("src/sub/solver_selection.jl", '#auto selection scip solver\n', "before",
     '# hip: there is no JuMP model; the design goes to libalmpc.so (see _model_predictive_control_design_hip in design_mpc.jl)\n'
     'function _JuMP_model_definition(method::AbstractImplementation, solver_selection::hip_solver_def)\n'
     '    error("mpc_solver = \\"hip\\" builds no JuMP model: it is taken in _model_predictive_control_design")\n'
     'end\n\n'),

    ("src/main/computation_mpc.jl", '    #force within the modeller\n', "before",
     '    # mpc_solver = "hip": x0 goes to the GPU (a vector of n values, or n * mpc_batch values for a batched controller)\n'
     '    if C.tuning.modeler isa AlmpcHIP.HipModeler\n'
     '        AlmpcHIP.update_initialization!(C.tuning.modeler, Vector{Float64}(initialization))\n'
     '        return\n'
     '    end\n\n'),

    ("src/main/computation_mpc.jl", '    #optimisation is done here\n', "before",
     '    # mpc_solver = "hip": one step on the GPU, results written into the same four matrices (for mpc_batch > 1 they were\n'
     '    # allocated with the instances side by side: n x (N+1)*batch is the memory of the n x (N+1) x batch array the library fills;\n'
     '    # AlmpcHIP.calculate! checks the lengths and throws DimensionMismatch rather than overrun them)\n'
     '    if C.tuning.modeler isa AlmpcHIP.HipModeler\n'
     '        AlmpcHIP.calculate!(\n'
     '            C.tuning.modeler,\n'
     '            C.computation_results.x,\n'
     '            C.computation_results.e_x,\n'
     '            C.computation_results.u,\n'
     '            C.computation_results.e_u,\n'
     '        )\n'
     '        return\n'
     '    end\n\n'),
]

# design_mpc.jl, linear-system method: branch right after the weights are made (first occurrence of the anchor)
LINEAR_BRANCH = '''    # mpc_solver = "hip": the whole design (prediction matrices, DARE, condensed QP) is built on the GPU behind the C ABI
    if mpc_solver isa hip_solver_def
        mpc_method_optimization isa LinearProgramming ||
            error("mpc_solver = \\"hip\\" on a linear system needs mpc_programming_type = \\"linear\\"")
        return _model_predictive_control_design_hip(
            system,
            horizon,
            sample_time,
            references,
            weights,
            mpc_terminal_ingredient,
            mpc_max_time;
            kws,
        )
    end

'''

# design_mpc.jl, black-box method: branch right after the model type is known
BLACKBOX_BRANCH = '''    # mpc_solver = "hip": Fnn models go to the GPU -- linear method: linearise as the LinearProgramming delegate does (first reference
    # for the dynamics, mpc_modeler_implementation_fnn.jl:38-46; last reference for P, _create_terminal_ingredient) and run the linear
    # design, or re-linearise every step on the device (mpc_linearization = "step"); non linear method: Gauss-Newton SQP on the device
    if mpc_solver isa hip_solver_def
        return _model_predictive_control_design_hip(
            mpc_method_optimization,
            model_type,
            system,
            horizon,
            sample_time,
            references,
            weights,
            mpc_terminal_ingredient,
            mpc_max_time;
            kws,
        )
    end

'''

APPENDIX = '''

# results of `batch` instances side by side: an n x (N+1)*batch Matrix is the memory of the n x (N+1) x batch array libalmpc.so writes
function _memory_allocation_initialization_results_hip(n::Int, m::Int, horizon::Int, batch::Int)
    initialization = Vector{Float64}(undef, n * batch)
    computation_results = ModelPredictiveControlResults(
        Array{Float64}(undef, n, (horizon + 1) * batch),
        Array{Float64}(undef, n, (horizon + 1) * batch),
        Array{Float64}(undef, m, horizon * batch),
        Array{Float64}(undef, m, horizon * batch),
    )
    return initialization, computation_results
end

"""
    _model_predictive_control_design_hip
Design of the linear model predictive control on the GPU (libalmpc.so through AlmpcHIP): same problem as the JuMP modeler of
mpc_modeler_implementation_linear.jl with the cost of _create_quadratic_cost_function and P = DARE. Extra keys: `mpc_batch`
(independent instances sharing this design, default 1: results are then allocated side by side, n x (N+1)*batch), `mpc_device`,
`mpc_rho_profile` ("scalar" | "stiffness"), `mpc_solver_options` (an `AlmpcHIP.AlmpcOpts`).
"""
function _model_predictive_control_design_hip(
    system::MathematicalSystems.ConstrainedLinearControlDiscreteSystem,
    horizon::Int,
    sample_time::Int,
    references::ReferencesStateInput,
    weights::WeightsCoefficient,
    terminal_ingredient::String,
    max_time;
    kws_...,
)

    # Get argument kws
    dict_kws = Dict{Symbol,Any}(kws_)
    kws = get(dict_kws, :kws, kws_)

    #get constraints of the dynamical system (low = last vertex, high = first vertex)
    x_hyperrectangle = LazySets.vertices_list(system.X)
    u_hyperrectangle = LazySets.vertices_list(system.U)
    x_constraints = hcat(x_hyperrectangle[end], x_hyperrectangle[begin])
    u_constraints = hcat(u_hyperrectangle[end], u_hyperrectangle[begin])

    state_box = haskey(kws, :mpc_state_constraint)
    batch = get(kws, :mpc_batch, 1)

    modeler = AlmpcHIP.design_hip(
        Matrix{Float64}(system.A),
        Matrix{Float64}(system.B),
        Matrix{Float64}(weights.Q),
        Matrix{Float64}(weights.R),
        Matrix{Float64}(weights.S),
        Vector{Float64}(u_constraints[:, 1]),
        Vector{Float64}(u_constraints[:, 2]),
        horizon;
        batch = batch,
        device = get(kws, :mpc_device, 0),
        x_ref = Matrix{Float64}(references.x),
        u_ref = Matrix{Float64}(references.u),
        opts = get(kws, :mpc_solver_options, AlmpcHIP.AlmpcOpts()),
        xmin = state_box ? Vector{Float64}(x_constraints[:, 1]) : nothing,
        xmax = state_box ? Vector{Float64}(x_constraints[:, 2]) : nothing,
        terminal = terminal_ingredient,
        rho_profile = get(kws, :mpc_rho_profile, "scalar"),
    )

    P_cost = AlmpcHIP.terminal_weight(modeler)

    tuning = ModelPredictiveControlTuning(
        modeler,
        references,
        horizon,
        weights,
        TerminalIngredient(terminal_ingredient, P_cost),
        sample_time,
        max_time,
    )

    initialization, computation_results = _memory_allocation_initialization_results_hip(
        size(system.A, 1),
        size(system.B, 2),
        horizon,
        batch,
    )

    return ModelPredictiveControlController(system, tuning, initialization, computation_results)
end

# activation code of libalmpc.so for the function get_activation_function returns (f[2][1].σ)
function _hip_activation_code(sigma)
    sigma === identity && return 0
    sigma === Flux.relu && return 1
    sigma === tanh && return 2
    (sigma === Flux.sigmoid || sigma === Flux.σ) && return 3
    sigma === Flux.swish && return 4
    error("mpc_solver = \\"hip\\": activation $(sigma) is not one of identity, relu, tanh, sigmoid, swish")
end

"""
    _model_predictive_control_design_hip
Black-box Fnn systems on the GPU. The weights are read from `Flux.params(system.f)` exactly as the NonLinearProgramming modeler does
(mpc_modeler_implementation_fnn.jl): first layer without bias, hidden layers with bias and activation, last layer without bias.
* LinearProgramming: Jacobians at `references.x[:, begin], references.u[:, begin]` (the LinearProgramming delegate) for the dynamics
  and at `[:, end]` for P (`_create_terminal_ingredient`), both by `AlmpcHIP.fnn_linearize`, then the linear design; with
  `mpc_linearization = "step"` every instance is re-linearised at its own state in every `calculate!` (`almpc_relin_fnn_*`).
* NonLinearProgramming: the same NLP the reference hands to Ipopt, by Gauss-Newton SQP on the device (`almpc_sqp_fnn_*`,
  `mpc_sqp_iterations` iterations per `calculate!`, default 20).
"""
function _model_predictive_control_design_hip(
    method::AbstractImplementation,
    model_type::AutomationLabsSystems.Fnn,
    system::MathematicalSystems.ConstrainedBlackBoxControlDiscreteSystem,
    horizon::Int,
    sample_time::Int,
    references::ReferencesStateInput,
    weights::WeightsCoefficient,
    terminal_ingredient::String,
    max_time;
    kws_...,
)

    # Get argument kws
    dict_kws = Dict{Symbol,Any}(kws_)
    kws = get(dict_kws, :kws, kws_)

    (method isa LinearProgramming || method isa NonLinearProgramming) ||
        error("mpc_solver = \\"hip\\" on a black-box system needs mpc_programming_type = \\"linear\\" or \\"non_linear\\"")

    #get constraints (low = last vertex, high = first vertex)
    x_hyperrectangle = LazySets.vertices_list(system.X)
    u_hyperrectangle = LazySets.vertices_list(system.U)
    x_constraints = hcat(x_hyperrectangle[end], x_hyperrectangle[begin])
    u_constraints = hcat(u_hyperrectangle[end], u_hyperrectangle[begin])
    umin = Vector{Float64}(u_constraints[:, 1])
    umax = Vector{Float64}(u_constraints[:, 2])
    state_box = haskey(kws, :mpc_state_constraint)
    xmin = state_box ? Vector{Float64}(x_constraints[:, 1]) : nothing
    xmax = state_box ? Vector{Float64}(x_constraints[:, 2]) : nothing

    #neural weights as the non linear modeler reads them
    nn_weights = Flux.params(system.f)
    nbr_neurons = size(nn_weights[1], 1)
    nbr_states = size(nn_weights[length(nn_weights)], 1)
    nbr_inputs_control = size(nn_weights[1], 2) - nbr_states
    nbr_hidden = trunc(Int, (length(nn_weights) - 2) / 2)
    W_in = Matrix{Float64}(nn_weights[1])
    W_out = Matrix{Float64}(nn_weights[length(nn_weights)])
    W_h = zeros(nbr_neurons, nbr_neurons, nbr_hidden)
    b_h = zeros(nbr_neurons, nbr_hidden)
    for (j, i) in enumerate(2:2:length(nn_weights)-1)
        W_h[:, :, j] = nn_weights[i]
        b_h[:, j] = nn_weights[i+1]
    end
    activation = _hip_activation_code(get_activation_function(model_type, system))

    batch = get(kws, :mpc_batch, 1)
    device = get(kws, :mpc_device, 0)
    opts = get(kws, :mpc_solver_options, AlmpcHIP.AlmpcOpts())
    x_ref = Matrix{Float64}(references.x)
    u_ref = Matrix{Float64}(references.u)
    Q = Matrix{Float64}(weights.Q)
    R = Matrix{Float64}(weights.R)
    S = Matrix{Float64}(weights.S)

    # dynamics at the first reference, terminal weight from the linearisation at the last one
    A_first, B_first, _ = AlmpcHIP.fnn_linearize(W_in, W_h, b_h, W_out, activation,
        reshape(x_ref[:, begin], :, 1), reshape(u_ref[:, begin], :, 1); device = device)
    A_last, B_last, _ = AlmpcHIP.fnn_linearize(W_in, W_h, b_h, W_out, activation,
        reshape(x_ref[:, end], :, 1), reshape(u_ref[:, end], :, 1); device = device)
    P_cost = AlmpcHIP.dare(A_last[:, :, 1], B_last[:, :, 1], Q, R)

    terminal_ingredient == "contractive" && error("terminal ingredient \\"contractive\\" is a quadratic constraint, not a QP row")
    modeler = AlmpcHIP.design_hip(
        A_first[:, :, 1],
        B_first[:, :, 1],
        Q,
        R,
        S,
        umin,
        umax,
        horizon;
        batch = batch,
        device = device,
        x_ref = x_ref,
        u_ref = u_ref,
        opts = opts,
        xmin = xmin,
        xmax = xmax,
        terminal = terminal_ingredient,
        rho_profile = get(kws, :mpc_rho_profile, "scalar"),
        P = P_cost,
    )
    if method isa NonLinearProgramming
        AlmpcHIP.set_state_rows!(modeler; xmin = xmin, xmax = xmax, terminal = terminal_ingredient)
        AlmpcHIP.design_sqp_fnn!(modeler, W_in, W_h, b_h, W_out, activation, Q, R, S, P_cost, umin, umax; x_ref = x_ref, u_ref = u_ref)
        modeler.mode = :sqp
        modeler.sqp_iterations = get(kws, :mpc_sqp_iterations, 20)
    elseif get(kws, :mpc_linearization, "reference") == "step"
        AlmpcHIP.set_state_rows!(modeler; xmin = xmin, xmax = xmax, terminal = terminal_ingredient)
        AlmpcHIP.design_relin_fnn!(modeler, W_in, W_h, b_h, W_out, activation, Q, R, S, P_cost, umin, umax; x_ref = x_ref, u_ref = u_ref)
        modeler.mode = :relin
    end

    tuning = ModelPredictiveControlTuning(
        modeler,
        references,
        horizon,
        weights,
        TerminalIngredient(terminal_ingredient, P_cost),
        sample_time,
        max_time,
    )

    initialization, computation_results =
        _memory_allocation_initialization_results_hip(nbr_states, nbr_inputs_control, horizon, batch)

    return ModelPredictiveControlController(system, tuning, initialization, computation_results)
end
'''


def insert(text, anchor, where, new, which=0):
    n = text.count(anchor)
    if n < which + 1:
        raise SystemExit(f"anchor not found ({which + 1}th occurrence): {anchor!r}")
    pos = -1
    for _ in range(which + 1):
        pos = text.index(anchor, pos + 1)
    if where == "before":
        return text[:pos] + new + text[pos:]
    return text[:pos + len(anchor)] + new + text[pos + len(anchor):]


def build(ref):
    tmp = tempfile.mkdtemp(prefix="almpc_patch_")
    try:
        for f in FILES:
            os.makedirs(os.path.dirname(os.path.join(tmp, f)), exist_ok=True)
            shutil.copy(os.path.join(ref, f), os.path.join(tmp, f))
        git = ["git", "-c", "user.email=x@x", "-c", "user.name=x", "-c", "core.autocrlf=false"]
        subprocess.check_call(["git", "init", "-q"], cwd=tmp)
        subprocess.check_call(git + ["add", "-A"], cwd=tmp)
        subprocess.check_call(git + ["commit", "-q", "-m", "reference"], cwd=tmp)
        for f, anchor, where, new in EDITS:
            p = os.path.join(tmp, f)
            t = open(p).read()
            if t.count(anchor) != 1:
                raise SystemExit(f"{f}: anchor must occur once, found {t.count(anchor)}: {anchor!r}")
            open(p, "w").write(insert(t, anchor, where, new))
        p = os.path.join(tmp, "src/sub/design_mpc.jl")
        t = open(p).read()
        # linear-system method: the first "# modeller implementation" block; black-box method: after the model type evaluation
        t = insert(t, "    # modeller implementation of model predictive control\n", "before", LINEAR_BRANCH)
        t = insert(t, "    model_type = AutomationLabsSystems.proceed_system_model_evaluation(system)\n\n", "after", BLACKBOX_BRANCH)
        t = t.rstrip("\n") + "\n" + APPENDIX
        open(p, "w").write(t)
        os.makedirs(os.path.join(tmp, "src/hip"), exist_ok=True)
        shutil.copy(os.path.join(HERE, "AlmpcHIP.jl"), os.path.join(tmp, "src/hip/AlmpcHIP.jl"))
        subprocess.check_call(git + ["add", "-A"], cwd=tmp)
        return subprocess.check_output(git + ["diff", "--cached", "--no-color"], cwd=tmp, text=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    patch = build(ref)
    with open(os.path.join(HERE, "reference_hip.patch"), "w") as f:
        f.write(patch)
    print(f"julia/reference_hip.patch: {len(patch.splitlines())} lines")
