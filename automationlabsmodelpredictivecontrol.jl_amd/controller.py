"""Host-side mirror of the reference's operator interface for the linear MPC path.

The reference is Julia; `julia` is not in this image, so the host side above the C ABI is Python that
keeps the reference's names, argument meaning and error behaviour (the Julia shim a maintainer would use
is julia/AlmpcHIP.jl).  Reference (paths relative to /root/reference):

  proceed_controller                     src/main/main_mpc.jl:22-53
  _design_reference_mpc                  src/main/main_mpc.jl:105-117
  _model_predictive_control_design       src/sub/design_mpc.jl:54-129   (ConstrainedLinearControlDiscreteSystem)
  _create_weights_coefficients           src/sub/design_mpc.jl:264-283
  _IMPLEMENTATION_SOLVER_LIST            src/sub/solver_selection.jl:9-14  (+ new tag "hip")
  update_initialization!                 src/main/computation_mpc.jl:17-29  -> update_initialization
  calculate!                             src/main/computation_mpc.jl:38-55  -> calculate
  structs                                src/types/types.jl:24-27,46-50,89-92,114-122,134-139,151-156

`_model_predictive_control_computation` is named by BASELINE.json but does not exist in the reference
(SURVEY.md section 0); here it is update_initialization + calculate for a batch.

Every solve goes through libalmpc.so (HIP); there is no CPU path in this module.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Optional

import numpy as np

from . import _capi
from .sharding import shard_range  # noqa: F401  (re-exported)

__all__ = [
    "Hyperrectangle", "ConstrainedLinearControlDiscreteSystem", "ConstrainedBlackBoxControlDiscreteSystem", "Fnn",
    "proceed_system_linearization", "ReferencesStateInput", "WeightsCoefficient",
    "TerminalIngredient", "ModelPredictiveControlTuning", "ModelPredictiveControlResults",
    "ModelPredictiveControlController", "proceed_controller", "_design_reference_mpc",
    "_model_predictive_control_design", "_create_weights_coefficients", "update_initialization", "calculate",
    "_model_predictive_control_computation", "HipModeler", "shard_range",
]


# ---- stand-ins for LazySets.Hyperrectangle / MathematicalSystems.ConstrainedLinearControlDiscreteSystem ----
@dataclasses.dataclass
class Hyperrectangle:
    low: np.ndarray
    high: np.ndarray

    def __post_init__(self):
        self.low = np.asarray(self.low, dtype=np.float64)
        self.high = np.asarray(self.high, dtype=np.float64)
        if self.low.shape != self.high.shape or np.any(self.low > self.high):
            raise ValueError("Hyperrectangle: need low <= high of equal length")


@dataclasses.dataclass
class ConstrainedLinearControlDiscreteSystem:
    A: np.ndarray
    B: np.ndarray
    X: Hyperrectangle
    U: Hyperrectangle

    def __post_init__(self):
        self.A = np.asarray(self.A, dtype=np.float64)
        self.B = np.asarray(self.B, dtype=np.float64)
        n, m = self.B.shape
        if self.A.shape != (n, n) or self.X.low.shape != (n,) or self.U.low.shape != (m,):
            raise ValueError("ConstrainedLinearControlDiscreteSystem: inconsistent dimensions")


@dataclasses.dataclass
class Fnn:
    """Feed-forward network in the layout the reference reads from Flux.params
    (src/sub/model_modeler_implementation/fnn/mpc_modeler_implementation_fnn.jl:88-107): W_in H x (n+m) (no bias, no
    activation), hidden layers (W_h[l], b_h[l]) with activation `act` ("identity" | "relu" | "tanh" | "sigmoid" | "swish"), W_out n x H (no bias).
    Also the model tag AutomationLabsSystems.Fnn() of src/sub/design_mpc.jl:176."""
    W_in: np.ndarray
    W_h: list
    b_h: list
    W_out: np.ndarray
    act: str = "relu"


@dataclasses.dataclass
class ConstrainedBlackBoxControlDiscreteSystem:
    """Stand-in for MathematicalSystems.ConstrainedBlackBoxControlDiscreteSystem(f, statedim, inputdim, X, U)."""
    f: Fnn
    statedim: int
    inputdim: int
    X: Hyperrectangle
    U: Hyperrectangle


def proceed_system_linearization(system: ConstrainedBlackBoxControlDiscreteSystem, state, input, device: int = 0):
    """AutomationLabsSystems.proceed_system_linearization(system, x, u) (call sites: .../fnn/...:42-46,
    src/sub/design_mpc.jl:319-326): the linear system (A, B) = Jacobians of f at (x, u), same constraint sets.
    Computed on the GPU (k_fnn_jacobian)."""
    f = system.f
    A, B = _capi.fnn_linearize(f.W_in, f.W_h, f.b_h, f.W_out, np.asarray(state, dtype=np.float64).reshape(1, -1),
                               np.asarray(input, dtype=np.float64).reshape(1, -1), act=f.act, device=device)
    return ConstrainedLinearControlDiscreteSystem(A[0], B[0], system.X, system.U)


# ---- structs of src/types/types.jl ------------------------------------------------------------------------
@dataclasses.dataclass
class ReferencesStateInput:
    x: np.ndarray  # n x (N+1)
    u: np.ndarray  # m x N


@dataclasses.dataclass
class WeightsCoefficient:
    Q: np.ndarray
    R: np.ndarray
    S: np.ndarray


@dataclasses.dataclass
class TerminalIngredient:
    Xf: Any
    P: np.ndarray


@dataclasses.dataclass
class ModelPredictiveControlTuning:
    modeler: Any  # the reference stores a JuMP model here (`modeler::Any`, types.jl:115); here a HipModeler
    reference: ReferencesStateInput
    horizon: int
    weights: WeightsCoefficient
    terminal_ingredient: TerminalIngredient
    sample_time: float
    max_time: int


@dataclasses.dataclass
class ModelPredictiveControlResults:
    x: np.ndarray
    e_x: np.ndarray
    u: np.ndarray
    e_u: np.ndarray


@dataclasses.dataclass
class ModelPredictiveControlController:
    system: Any
    tuning: ModelPredictiveControlTuning
    initialization: np.ndarray
    computation_results: ModelPredictiveControlResults


_DEFAULT_PARAMETERS_MODEL_PREDICTIVE_CONTROL = dict(  # src/main/main_mpc.jl:87-94
    mpc_solver="auto", mpc_terminal_ingredient="none", mpc_Q=100.0, mpc_R=0.1, mpc_S=0.0, mpc_max_time=30.0)

# src/sub/solver_selection.jl:9-14 plus the new "hip" tag; "auto" for the linear method resolves to "hip" here
# (the reference resolves it to SCIP, solver_selection.jl:56-65); the CPU solvers are not part of this build.
_IMPLEMENTATION_SOLVER_LIST = ("osqp", "scip", "ipopt", "auto", "hip")
IMPLEMENTATION_PROGRAMMING_LIST = ("linear", "non_linear", "mixed_linear", "fuzzy_linear")  # src/types/types.jl:229-234


def _kws(kws_):
    # the reference's idiom: dict_kws = Dict(kws_); kws = get(dict_kws, :kws, kws_)  (src/main/main_mpc.jl:33-34)
    return dict(kws_.get("kws", kws_))


class HipModeler:
    """What sits in `tuning.modeler` instead of a JuMP model: the almpc handle plus solver options."""

    def __init__(self, solver: _capi.Solver, opts: _capi.almpc_opts, batch: int):
        self.solver, self.opts, self.batch = solver, opts, batch
        self.relinearize = None  # dict for mpc_linearization='step' (black-box models), see _design_blackbox
        self.sqp = None          # dict for mpc_programming_type='non_linear', see _design_blackbox_nonlinear
        self.allow_unsolved = False  # kw mpc_allow_unsolved: calculate! returns the iterate of an instance without a certificate instead of raising


def _design_reference_mpc(state_reference, input_reference, horizon: int) -> ReferencesStateInput:
    xr = np.asarray(state_reference, dtype=np.float64).reshape(-1, 1)
    ur = np.asarray(input_reference, dtype=np.float64).reshape(-1, 1)
    return ReferencesStateInput(xr * np.ones((xr.shape[0], horizon + 1)), ur * np.ones((ur.shape[0], horizon)))


def _create_weights_coefficients(system: ConstrainedLinearControlDiscreteSystem, **kws_) -> WeightsCoefficient:
    kws = _kws(kws_)
    D = _DEFAULT_PARAMETERS_MODEL_PREDICTIVE_CONTROL
    n, m = system.B.shape
    return WeightsCoefficient(float(kws.get("mpc_Q", D["mpc_Q"])) * np.eye(n),
                              float(kws.get("mpc_R", D["mpc_R"])) * np.eye(m),
                              float(kws.get("mpc_S", D["mpc_S"])) * np.eye(m))


def proceed_controller(system, mpc_controller_type: str, mpc_horizon: int, mpc_sample_time: int,
                       mpc_state_reference, mpc_input_reference, **kws_):
    """Same positional contract as the reference (src/main/main_mpc.jl:22-30).  Returns None for a
    controller type other than "model_predictive_control", as the reference falls through."""
    kws = _kws(kws_)
    if not isinstance(mpc_horizon, (int, np.integer)) or not isinstance(mpc_sample_time, (int, np.integer)):
        raise TypeError("mpc_horizon and mpc_sample_time must be Int (src/main/main_mpc.jl:25-26)")
    if mpc_controller_type == "model_predictive_control":
        refs = _design_reference_mpc(mpc_state_reference, mpc_input_reference, mpc_horizon)
        return _model_predictive_control_design(system, mpc_horizon, mpc_sample_time, refs, kws=kws)
    return None


def _model_predictive_control_design(system, horizon: int, sample_time: int, references: ReferencesStateInput, **kws_):
    if isinstance(system, ConstrainedBlackBoxControlDiscreteSystem):
        return _design_blackbox(system, horizon, sample_time, references, **kws_)
    return _design_linear(system, horizon, sample_time, references, **kws_)


def _design_blackbox(system: ConstrainedBlackBoxControlDiscreteSystem, horizon: int, sample_time: int,
                     references: ReferencesStateInput, **kws_):
    """Black-box (Fnn) model, LinearProgramming branch (src/sub/design_mpc.jl:143-225 ->
    .../fnn/mpc_modeler_implementation_fnn.jl:23-58): dynamics linearised at the FIRST reference, terminal weight
    P = DARE at the linearisation about the LAST reference (src/sub/design_mpc.jl:312-327), then the linear path.

    Extension of this build (BASELINE.json configs[3]), kw mpc_linearization = "step": the dynamics are re-linearised at every
    instance's own current state (and the first input reference) in update_initialization!, each instance gets the
    reference's QP for its own (A_i, B_i) (almpc_design_batched); P stays the design-time terminal weight.  The default
    "reference" is the reference's behaviour: one linearisation at design time, shared by the batch."""
    kws = _kws(kws_)
    lin_mode = kws.get("mpc_linearization", "reference")
    if lin_mode not in ("reference", "step"):
        raise ValueError("mpc_linearization must be 'reference' or 'step'")
    if not isinstance(system.f, Fnn):
        raise NotImplementedError("only the Fnn model family is built (SURVEY.md section 2, components 8-14 are out of scope)")
    dev = int(kws.get("mpc_device", 0))
    x_ref, u_ref = np.asarray(references.x, dtype=np.float64), np.asarray(references.u, dtype=np.float64)
    lin_first = proceed_system_linearization(system, x_ref[:, 0], u_ref[:, 0], device=dev)
    lin_last = proceed_system_linearization(system, x_ref[:, -1], u_ref[:, -1], device=dev)
    weights = _create_weights_coefficients(lin_first, kws=kws)
    P = _capi.dare(lin_last.A, lin_last.B, weights.Q, weights.R)
    if kws.get("mpc_programming_type", "linear") == "non_linear":
        return _design_blackbox_nonlinear(system, horizon, sample_time, references, weights, P, kws)
    C = _design_linear(lin_first, horizon, sample_time, references, kws=kws, _terminal_P=P)
    C.system = system
    if lin_mode == "step":
        state_box = "mpc_state_constraint" in kws   # the rows of .../fnn/mpc_modeler_implementation_fnn.jl:52-58, per instance
        term_eq = kws.get("mpc_terminal_ingredient", "none") == "equality"
        mod = C.tuning.modeler
        sopt = dict(kws.get("mpc_solver_options", {}))
        f = system.f
        # instances whose linearisation is open-loop unstable (condensed Hessian singular to working precision) are redone in the
        # multiple-shooting form by the stage-wise solvers: the library's default; mpc_structured_fallback = False switches it off
        if not kws.get("mpc_structured_fallback", True):
            mod.solver._check(mod.solver.L.almpc_set_structured_fallback(mod.solver.h, 0))
        # device-resident pipeline (almpc_relin_fnn_*): Jacobians -> per-instance designs -> step, no host pointers per step
        mod.solver.relin_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, references.x, references.u, weights.Q, weights.R, weights.S, np.array(P),
                                   system.U.low, system.U.high, act=f.act, rho=float(sopt.get("rho", 0.1)),
                                   sigma=float(sopt.get("sigma", 1e-6)), rho_profile=kws.get("mpc_rho_profile", "scalar"),
                                   xmin=system.X.low if state_box else None, xmax=system.X.high if state_box else None,
                                   terminal="equality" if term_eq else "none")
        mod.relinearize = dict(system=system, device=dev)
    return C


def _design_blackbox_nonlinear(system, horizon, sample_time, references, weights, P, kws):
    """Black-box (Fnn) model, NonLinearProgramming branch (.../fnn/mpc_modeler_implementation_fnn.jl:73-189): the network itself
    is the equality constraint x[:,k+1] = fnn(x[:,k], u[:,k]) and the reference gives the NLP to Ipopt
    (src/sub/solver_selection.jl:100-104).  Here the same NLP goes through the device-resident SQP loop (almpc_sqp_fnn_*).
    Keys of this build: mpc_sqp_iterations (outer iterations per calculate!, default 10), mpc_sqp_step (step length, default 1),
    mpc_sqp_step_rule ("merit": steps safeguarded by the l1 merit function, default; "fixed"),
    mpc_sqp_warm_start (start each calculate! from the previous inputs shifted by one stage, default True)."""
    D = _DEFAULT_PARAMETERS_MODEL_PREDICTIVE_CONTROL
    solver_name = kws.get("mpc_solver", D["mpc_solver"])
    if solver_name not in _IMPLEMENTATION_SOLVER_LIST:
        raise KeyError(solver_name)
    if solver_name in ("osqp", "scip", "ipopt"):
        raise NotImplementedError(f"mpc_solver={solver_name!r} is the reference's CPU path; this build provides 'hip' (and 'auto' -> 'hip')")
    terminal = kws.get("mpc_terminal_ingredient", D["mpc_terminal_ingredient"])
    if terminal == "contractive":
        raise NotImplementedError("terminal ingredient 'contractive' is a quadratic constraint (src/sub/design_mpc.jl:333-340), not a QP row")
    state_box = "mpc_state_constraint" in kws       # .../fnn/mpc_modeler_implementation_fnn.jl:146-153: rows of every SQP iteration's QP
    f = system.f
    n, m = system.statedim, system.inputdim
    batch = int(kws.get("mpc_batch", 1))
    x_ref, u_ref = np.asarray(references.x, dtype=np.float64), np.asarray(references.u, dtype=np.float64)
    if x_ref.shape != (n, horizon + 1) or u_ref.shape != (m, horizon):
        raise ValueError("references must be n x (N+1) and m x N")
    sopt = dict(kws.get("mpc_solver_options", {}))
    solver = _capi.Solver(n, m, horizon, batch, device=int(kws.get("mpc_device", 0)), timing=bool(kws.get("mpc_timing", False)))
    solver.sqp_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, weights.Q, weights.R, weights.S, P, system.U.low, system.U.high,
                         act=f.act, rho=float(sopt.get("rho", 0.1)), sigma=float(sopt.get("sigma", 1e-6)),
                         rho_profile=kws.get("mpc_rho_profile", "scalar"),
                         xmin=system.X.low if state_box else None, xmax=system.X.high if state_box else None,
                         terminal="equality" if terminal == "equality" else "none",
                         qp_solver=kws.get("mpc_sqp_qp_solver", "condensed"))   # "structured": every QP through k_riccati
    mod = HipModeler(solver, _capi.default_opts(**sopt), batch)
    mod.allow_unsolved = bool(kws.get("mpc_allow_unsolved", False))
    mod.sqp = dict(iterations=int(kws.get("mpc_sqp_iterations", 10)), step=float(kws.get("mpc_sqp_step", 1.0)),
                   warm_start=bool(kws.get("mpc_sqp_warm_start", True)), u_prev=None,
                   step_rule=kws.get("mpc_sqp_step_rule", "merit"))
    tuning = ModelPredictiveControlTuning(mod, references, horizon, weights, TerminalIngredient(terminal, np.array(P)),
                                          float(sample_time), int(kws.get("mpc_max_time", D["mpc_max_time"])))
    shape = (lambda *s: s) if batch == 1 else (lambda *s: (batch, *s))
    results = ModelPredictiveControlResults(np.empty(shape(n, horizon + 1)), np.empty(shape(n, horizon + 1)),
                                            np.empty(shape(m, horizon)), np.empty(shape(m, horizon)))
    return ModelPredictiveControlController(system, tuning, np.empty(shape(n)), results)


def _design_linear(system: ConstrainedLinearControlDiscreteSystem, horizon: int, sample_time: int,
                   references: ReferencesStateInput, _terminal_P=None, **kws_):
    """Design for the discrete linear system (src/sub/design_mpc.jl:54-129).  Extra keys of this build:
    mpc_batch (instances sharing this design, default 1), mpc_device (HIP device id, default 0),
    mpc_solver_options (dict of almpc_opts fields), mpc_timing (bool)."""
    kws = _kws(kws_)
    D = _DEFAULT_PARAMETERS_MODEL_PREDICTIVE_CONTROL
    ptype = kws.get("mpc_programming_type", "linear")
    if ptype not in IMPLEMENTATION_PROGRAMMING_LIST:
        raise KeyError(ptype)  # the reference indexes a NamedTuple and throws
    if ptype != "linear":
        raise NotImplementedError(f"mpc_programming_type={ptype!r}: only the LinearProgramming path is built (SURVEY.md section 8)")
    solver_name = kws.get("mpc_solver", D["mpc_solver"])
    if solver_name not in _IMPLEMENTATION_SOLVER_LIST:
        raise KeyError(solver_name)
    if solver_name in ("osqp", "scip", "ipopt"):
        raise NotImplementedError(f"mpc_solver={solver_name!r} is the reference's CPU path; this build provides 'hip' (and 'auto' -> 'hip')")
    terminal = kws.get("mpc_terminal_ingredient", D["mpc_terminal_ingredient"])
    if terminal == "contractive":
        raise NotImplementedError("terminal ingredient 'contractive' is a quadratic constraint (src/sub/design_mpc.jl:333-340), "
                                  "not a QP row: the reference itself cannot pass it to OSQP")
    if terminal == "neighborhood":  # the reference only warns and adds nothing (src/sub/design_mpc.jl:342-345)
        import warnings
        warnings.warn("neighborhood terminal state constraint is not yet implemented")
    elif terminal not in ("none", "equality"):
        # the reference's final `else` (src/sub/design_mpc.jl:389-391): an unknown string adds no terminal constraint; it is stored as given
        # in TerminalIngredient.Xf there, and here
        pass
    # the state box exists only if the kw is PRESENT (its value is never read), bounds come from system.X (..linear.jl:62-70)
    state_box = "mpc_state_constraint" in kws
    max_time = kws.get("mpc_max_time", D["mpc_max_time"])
    weights = _create_weights_coefficients(system, kws=kws)
    n, m = system.B.shape
    batch = int(kws.get("mpc_batch", 1))
    x_ref, u_ref = np.asarray(references.x, dtype=np.float64), np.asarray(references.u, dtype=np.float64)
    if x_ref.shape != (n, horizon + 1) or u_ref.shape != (m, horizon):
        raise ValueError("references must be n x (N+1) and m x N")
    sopt = dict(kws.get("mpc_solver_options", {}))
    rho, sigma = float(sopt.get("rho", 0.1)), float(sopt.get("sigma", 1e-6))
    solver = _capi.Solver(n, m, horizon, batch, device=int(kws.get("mpc_device", 0)), timing=bool(kws.get("mpc_timing", False)))
    # bounds as the reference reads them: low = last vertex, high = first vertex of the hyperrectangle
    # (..linear.jl:34-38); P = DARE at the (linear) system (src/sub/design_mpc.jl:327), computed in the library.
    solver.design_shared(system.A, system.B, weights.Q, weights.R, weights.S, _terminal_P, system.U.low, system.U.high,
                         xmin=system.X.low if state_box else None, xmax=system.X.high if state_box else None,
                         rho=rho, sigma=sigma, terminal="equality" if terminal == "equality" else "none",
                         rho_profile=kws.get("mpc_rho_profile", "scalar"))
    solver.set_reference(x_ref, u_ref)
    P = solver.get_design()["P"]
    opts = _capi.default_opts(**sopt)
    modeler_ = HipModeler(solver, opts, batch)
    modeler_.allow_unsolved = bool(kws.get("mpc_allow_unsolved", False))
    tuning = ModelPredictiveControlTuning(modeler_, references, horizon, weights,
                                          TerminalIngredient(terminal, np.array(P)), float(sample_time), int(max_time))
    shape = (lambda *s: s) if batch == 1 else (lambda *s: (batch, *s))
    results = ModelPredictiveControlResults(np.empty(shape(n, horizon + 1)), np.empty(shape(n, horizon + 1)),
                                            np.empty(shape(m, horizon)), np.empty(shape(m, horizon)))
    return ModelPredictiveControlController(system, tuning, np.empty(shape(n)), results)


def update_initialization(C: ModelPredictiveControlController, initialization) -> None:
    """update_initialization!(C, x0): x0 of length n (batch 1) or shape (batch, n)."""
    mod: HipModeler = C.tuning.modeler
    x0 = np.asarray(initialization, dtype=np.float64)
    n = C.tuning.modeler.solver.n
    if x0.size != mod.batch * n:
        raise ValueError(f"initialization must hold {mod.batch} x {n} values")
    C.initialization = x0.reshape((n,) if mod.batch == 1 else (mod.batch, n)).copy()
    if getattr(mod, "sqp", None) is not None:  # non_linear: the SQP iterate restarts from the network's own rollout
        up = mod.sqp["u_prev"] if mod.sqp["warm_start"] else None
        ug = None if up is None else np.concatenate([up[:, :, 1:], up[:, :, -1:]], axis=2)
        mod.solver.sqp_fnn_start(x0.reshape(mod.batch, n), ug)
        return
    # mpc_linearization='step': the Jacobians at (x0_i, u_ref[:,1]) and the per-instance designs are part of calculate! (on the device)
    mod.solver.update_initialization(x0.reshape(mod.batch, n))


def calculate(C: ModelPredictiveControlController) -> None:
    """calculate!(C): solve and copy u, e_u, x, e_x into C.computation_results.  The reference does not
    check the solver status and lets JuMP.value throw when no solution exists; here a non-finite instance
    raises ArithmeticError, and per-instance status/iterations are kept on the modeler."""
    mod: HipModeler = C.tuning.modeler
    if getattr(mod, "sqp", None) is not None:
        mod.last_sqp_history = mod.solver.sqp_fnn_iterate(mod.sqp["iterations"], mod.sqp["step"], mod.opts, step_rule=mod.sqp["step_rule"])
    elif getattr(mod, "relinearize", None) is not None:
        mod.solver.relin_fnn_step(mod.opts)
    else:
        mod.solver.calculate(mod.opts)
    r = mod.solver.get_results()
    mod.last_status, mod.last_iters, mod.last_polish_iters = r["status"], r["iters"], r["polish_iters"]
    if np.any(r["status"] == _capi.NON_FINITE):
        raise ArithmeticError("calculate!: non-finite values in at least one instance (no solution to read)")
    if np.any(r["status"] == _capi.INFEASIBLE):  # the reference: JuMP.value throws when the solver has no primal
        raise ArithmeticError("calculate!: infeasible problem in at least one instance (state box / terminal equality)")
    # status 1 with the exact finish on: no certificate even after the library's stage-wise redo (an iteration cap, a working set
    # beyond 128 rows).  The reference returns a solution or throws: raise, unless kw mpc_allow_unsolved = True asked for the iterate
    # (with polish off status 1 is OSQP's ITERATION_LIMIT, with which JuMP.value still returns)
    if np.any(r["status"] == 1) and int(mod.opts.polish) != 0 and not getattr(mod, "allow_unsolved", False):
        raise ArithmeticError(f"calculate!: {int((r['status'] == 1).sum())} instance(s) without an optimality certificate "
                              "(mpc_allow_unsolved = True returns the iterate; modeler.last_status says which)")
    res = C.computation_results
    for k in ("x", "e_x", "u", "e_u"):
        getattr(res, k)[...] = r[k][0] if mod.batch == 1 else r[k]
    if getattr(mod, "sqp", None) is not None:
        mod.sqp["u_prev"] = r["u"].copy()


def _model_predictive_control_computation(C: ModelPredictiveControlController, X0):
    """Batch step: update_initialization! + calculate!; returns C.computation_results."""
    update_initialization(C, X0)
    calculate(C)
    return C.computation_results
