# ref_osqp_bench.jl -- times the REFERENCE's own CPU path on the benchmark workload, for a box that has Julia and the
# reference's packages (this image has neither: SURVEY.md section 8c/8d-iii).  Not used by bench.py; the figure it prints goes
# beside `cpu_baseline` by hand.
#
#   julia --project=<checkout of AutomationLabsModelPredictiveControl.jl> julia/ref_osqp_bench.jl [instances] [amplitude]
#
# What it does, with the reference's API exactly as its tests use it (test/computation_mpc_test.jl:981-1054):
#   proceed_controller(sys, "model_predictive_control", 30, 1, x_ref, u_ref; mpc_solver = "osqp")   src/main/main_mpc.jl:22-53
#   update_initialization!(C, x0); calculate!(C)                                                    src/main/computation_mpc.jl:17-55
# on the hover-linearised quadrotor of BASELINE configs[1] (exact ZOH, Ts = 0.1), one instance after the other (the
# reference has no batch API), initial states from the same SplitMix64 -> Box-Muller generator as oracle/mpc_oracle.py
# (seed 0x5EED0002, stream = instance index).  Output: one JSON line {instance_steps_per_s, batch_steps_per_s_4096, threads}.
using LinearAlgebra
import AutomationLabsModelPredictiveControl: proceed_controller, update_initialization!, calculate!
import MathematicalSystems, LazySets

mix64(z::UInt64) = begin
    z = (z ⊻ (z >> 30)) * 0xBF58476D1CE4E5B9
    z = (z ⊻ (z >> 27)) * 0x94D049BB133111EB
    z ⊻ (z >> 31)
end

# normals of one stream: the generator of oracle/mpc_oracle.py::splitmix_normal
function splitmix_normal(seed::UInt64, instance::Int, dim::Int)
    state = mix64(seed + 0x632BE59BD9B4E019 * UInt64(instance + 1))
    out = zeros(dim)
    i = 1
    while i <= dim
        state += 0x9E3779B97F4A7C15; u1 = Float64(mix64(state) >> 11) * 2.0^-53
        state += 0x9E3779B97F4A7C15; u2 = Float64(mix64(state) >> 11) * 2.0^-53
        r = sqrt(-2.0 * log(1.0 - u1))
        out[i] = r * cos(2pi * u2)
        if i + 1 <= dim
            out[i+1] = r * sin(2pi * u2)
        end
        i += 2
    end
    out
end

function quadrotor(Ts = 0.1; mass = 0.5, J = (4e-3, 4e-3, 8e-3), g = 9.81)
    Ac = zeros(12, 12); Bc = zeros(12, 4)
    Ac[1:3, 4:6] = I(3); Ac[4, 8] = g; Ac[5, 7] = -g; Ac[7:9, 10:12] = I(3)
    Bc[6, 1] = 1 / mass; Bc[10, 2] = 1 / J[1]; Bc[11, 3] = 1 / J[2]; Bc[12, 4] = 1 / J[3]
    M = exp([Ac Bc; zeros(4, 16)] * Ts)
    M[1:12, 1:12], M[1:12, 13:16]
end

function main()
    n_inst = length(ARGS) >= 1 ? parse(Int, ARGS[1]) : 256
    amp = length(ARGS) >= 2 ? parse(Float64, ARGS[2]) : 1.0
    A, B = quadrotor()
    X = LazySets.Hyperrectangle(low = fill(-1e3, 12), high = fill(1e3, 12))
    U = LazySets.Hyperrectangle(low = [-2.0, -0.05, -0.05, -0.02], high = [3.0, 0.05, 0.05, 0.02])
    sys = MathematicalSystems.ConstrainedLinearControlDiscreteSystem(A, B, X, U)
    C = proceed_controller(sys, "model_predictive_control", 30, 1, zeros(12), zeros(4); mpc_solver = "osqp")
    scale = [1, 1, 1, 0.5, 0.5, 0.5, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1]
    x0s = [amp .* scale .* splitmix_normal(UInt64(0x5EED0002), i - 1, 12) for i in 1:n_inst]
    update_initialization!(C, x0s[1]); calculate!(C)            # compile + OSQP setup
    t0 = time()
    for x0 in x0s
        update_initialization!(C, x0)
        calculate!(C)
    end
    el = time() - t0
    println("{\"instance_steps_per_s\": ", n_inst / el, ", \"batch_steps_per_s_4096\": ", n_inst / el / 4096,
            ", \"threads\": 1, \"instances\": ", n_inst, ", \"amplitude\": ", amp, ", \"solver\": \"osqp (reference defaults)\"}")
end

main()
