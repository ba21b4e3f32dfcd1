"""ctypes binding of libalmpc.so (include/almpc.h).  The HIP library is the only compute path: if it is
missing or fails to load this module raises -- there is no CPU fallback."""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ALMPC_LIB overrides the library path (A/B builds of the same ABI); the default is the in-tree build
LIB_PATH = os.environ.get("ALMPC_LIB") or os.path.join(_HERE, "lib", "libalmpc.so")

ALMPC_OK = 0
ERR_NAMES = {0: "ALMPC_OK", -1: "ALMPC_ERR_INVALID", -2: "ALMPC_ERR_NO_DEVICE", -3: "ALMPC_ERR_HIP",
             -4: "ALMPC_ERR_UNSUPPORTED", -5: "ALMPC_ERR_NOT_DESIGNED", -6: "ALMPC_ERR_NUMERIC"}
FLAG_TIMING = 0x1
FLAG_STRUCTURED = 0x2
SOLVED, MAX_ITER, NON_FINITE, INFEASIBLE = 0, 1, 2, 3


class AlmpcError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")


class almpc_opts(ctypes.Structure):
    _fields_ = [("rho", ctypes.c_double), ("sigma", ctypes.c_double), ("alpha", ctypes.c_double),
                ("eps_abs", ctypes.c_double), ("eps_rel", ctypes.c_double), ("max_iter", ctypes.c_int32),
                ("check_every", ctypes.c_int32), ("polish", ctypes.c_int32), ("polish_max_iter", ctypes.c_int32),
                ("warm_start", ctypes.c_int32), ("reserved", ctypes.c_int32 * 3)]


_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_fp = ctypes.POINTER(ctypes.c_float)
_hp = ctypes.c_void_p
_lib = None


def load():
    """Load libalmpc.so and declare the prototypes of include/almpc.h.  Raises OSError if the
    library has not been built (`make lib` or `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} not found: build the HIP library first (make lib). There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    L.almpc_default_opts.argtypes = [ctypes.POINTER(almpc_opts)]
    L.almpc_default_opts.restype = None
    L.almpc_create.argtypes = [ctypes.POINTER(_hp), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_int, ctypes.c_uint32]
    L.almpc_destroy.argtypes = [_hp]
    L.almpc_destroy.restype = None
    L.almpc_last_error.argtypes = [_hp]
    L.almpc_last_error.restype = ctypes.c_char_p
    L.almpc_design_shared.argtypes = [_hp] + [_dp] * 10 + [ctypes.c_double, ctypes.c_double]
    L.almpc_design_batched.argtypes = [_hp] + [_dp] * 6 + [ctypes.c_int, _dp, _dp, ctypes.c_double, ctypes.c_double]
    L.almpc_design_batched.restype = ctypes.c_int
    L.almpc_get_design_instance.argtypes = [_hp, ctypes.c_int, _dp, _dp, _dp]
    L.almpc_design_ltv.argtypes = [_hp] + [_dp] * 11 + [ctypes.c_int, _dp, _dp, ctypes.c_double, ctypes.c_double]
    L.almpc_design_ltv.restype = ctypes.c_int
    L.almpc_get_gradient_instance.argtypes = [_hp, ctypes.c_int, _dp]
    L.almpc_get_gradient_instance.restype = ctypes.c_int
    L.almpc_sqp_fnn_setup.argtypes = [_hp, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [_dp] * 10 + [ctypes.c_int, _dp, _dp,
                                                                                                     ctypes.c_double, ctypes.c_double]
    L.almpc_sqp_fnn_setup.restype = ctypes.c_int
    L.almpc_sqp_fnn_start.argtypes = [_hp, _dp, _dp]
    L.almpc_sqp_fnn_start.restype = ctypes.c_int
    L.almpc_sqp_fnn_iterate.argtypes = [_hp, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, _dp, _dp]
    L.almpc_sqp_fnn_iterate.restype = ctypes.c_int
    L.almpc_sqp_fnn_skipped.argtypes = [_hp, _ip]
    L.almpc_sqp_fnn_set_step_rule.argtypes = [_hp, ctypes.c_int]
    L.almpc_sqp_fnn_set_step_rule.restype = ctypes.c_int
    L.almpc_sqp_fnn_skipped.restype = ctypes.c_int
    L.almpc_get_design_instance.restype = ctypes.c_int
    L.almpc_set_reference.argtypes = [_hp, _dp, _dp, ctypes.c_int]
    L.almpc_set_terminal_equality.argtypes = [_hp, ctypes.c_int]
    L.almpc_set_state_box.argtypes = [_hp, _dp, _dp]
    L.almpc_set_state_box.restype = ctypes.c_int
    L.almpc_sqp_fnn_set_structured.argtypes = [_hp, ctypes.c_int]
    L.almpc_sqp_fnn_set_structured.restype = ctypes.c_int
    L.almpc_set_rho_profile.argtypes = [_hp, ctypes.c_int]
    L.almpc_set_rho_profile.restype = ctypes.c_int
    L.almpc_set_step_fusion.argtypes = [_hp, ctypes.c_int]
    L.almpc_set_step_fusion.restype = ctypes.c_int
    L.almpc_set_terminal_equality.restype = ctypes.c_int
    L.almpc_update_initialization.argtypes = [_hp, _dp]
    L.almpc_update_initialization_device.argtypes = [_hp, ctypes.c_void_p]
    L.almpc_calculate.argtypes = [_hp, ctypes.POINTER(almpc_opts)]
    L.almpc_calculate_async.argtypes = [_hp, ctypes.POINTER(almpc_opts)]
    L.almpc_synchronize.argtypes = [_hp]
    L.almpc_get_results.argtypes = [_hp, _dp, _dp, _dp, _dp, _ip, _ip, _ip]
    L.almpc_get_design.argtypes = [_hp, _dp, _dp, _dp, _dp]
    L.almpc_device_results.argtypes = [_hp] + [ctypes.POINTER(ctypes.c_void_p)] * 4
    L.almpc_get_timing.argtypes = [_hp, _fp, _fp, _fp, _fp]
    L.almpc_timing_reset.argtypes = [_hp, ctypes.c_int]
    L.almpc_timing_set_stride.argtypes = [_hp, ctypes.c_int]
    L.almpc_timing_set_stride.restype = ctypes.c_int
    L.almpc_debug_poison_lds.argtypes = [_hp]
    L.almpc_advance_plant.argtypes = [_hp]
    L.almpc_advance_plant.restype = ctypes.c_int
    L.almpc_dare.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp, _dp]
    L.almpc_dare.restype = ctypes.c_int
    L.almpc_set_structured_fallback.argtypes = [_hp, ctypes.c_int]
    L.almpc_relin_fnn_setup.argtypes = [_hp, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [_dp] * 12 + [ctypes.c_double, ctypes.c_double]
    L.almpc_relin_fnn_setup.restype = ctypes.c_int
    L.almpc_relin_fnn_step.argtypes = [_hp, ctypes.POINTER(almpc_opts)]
    L.almpc_relin_fnn_step_async.argtypes = [_hp, ctypes.POINTER(almpc_opts)]
    L.almpc_relin_fnn_timing.argtypes = [_hp, _fp, _fp, _fp]
    L.almpc_relin_fnn_advance.argtypes = [_hp]
    L.almpc_comm_unique_id.argtypes = [ctypes.c_char_p]
    L.almpc_comm_init.argtypes = [_hp, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    L.almpc_comm_summary.argtypes = [_hp, ctypes.POINTER(ctypes.c_int64)]
    L.almpc_comm_allgather_first_input.argtypes = [_hp, _dp, ctypes.POINTER(_dp)]
    L.almpc_fnn_linearize.argtypes = [ctypes.c_int] * 6 + [_dp] * 4 + [ctypes.c_int] + [_dp] * 5
    L.almpc_fnn_linearize.restype = ctypes.c_int
    L.almpc_debug_poison_lds.restype = ctypes.c_int
    L.almpc_timing_summary.argtypes = [_hp, ctypes.POINTER(ctypes.c_int)] + [_dp] * 4
    # host-facing step path (pinned staging, copy streams) and one-process multi-GPU groups
    L.almpc_update_initialization_async.argtypes = [_hp, _dp]
    L.almpc_x0_staging.argtypes = [_hp, ctypes.POINTER(_dp)]
    L.almpc_get_results_async.argtypes = [_hp, ctypes.c_uint32]
    L.almpc_get_results_wait.argtypes = [_hp, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _ip]
    L.almpc_host_results.argtypes = [_hp, ctypes.c_int] + [ctypes.POINTER(ctypes.c_void_p)] * 8
    L.almpc_get_first_input.argtypes = [_hp, _dp]
    L.almpc_group_create.argtypes = [ctypes.POINTER(_hp), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_int), ctypes.c_uint32]
    L.almpc_group_destroy.argtypes = [_hp]
    L.almpc_group_destroy.restype = None
    L.almpc_group_last_error.argtypes = [_hp]
    L.almpc_group_last_error.restype = ctypes.c_char_p
    L.almpc_group_size.argtypes = [_hp]
    L.almpc_group_handle.argtypes = [_hp, ctypes.c_int]
    L.almpc_group_handle.restype = _hp
    L.almpc_group_shard.argtypes = [_hp, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.almpc_group_design_shared.argtypes = [_hp] + [_dp] * 10 + [ctypes.c_double, ctypes.c_double]
    L.almpc_group_set_reference.argtypes = [_hp, _dp, _dp, ctypes.c_int]
    L.almpc_group_update_initialization.argtypes = [_hp, _dp]
    L.almpc_group_calculate.argtypes = [_hp, ctypes.POINTER(almpc_opts)]
    L.almpc_group_calculate_async.argtypes = [_hp, ctypes.POINTER(almpc_opts)]
    L.almpc_group_synchronize.argtypes = [_hp]
    L.almpc_group_get_results.argtypes = [_hp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _ip]
    # group forms of everything a handle can do
    for nm_ in ("almpc_group_set_terminal_equality", "almpc_group_set_rho_profile", "almpc_group_set_structured_fallback",
                "almpc_group_sqp_fnn_set_structured", "almpc_group_sqp_fnn_set_step_rule"):
        getattr(L, nm_).argtypes = [_hp, ctypes.c_int]
    L.almpc_group_set_state_box.argtypes = [_hp, _dp, _dp]
    L.almpc_group_design_batched.argtypes = [_hp] + [_dp] * 6 + [ctypes.c_int, _dp, _dp, ctypes.c_double, ctypes.c_double]
    L.almpc_group_relin_fnn_setup.argtypes = [_hp, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [_dp] * 12 + [ctypes.c_double, ctypes.c_double]
    L.almpc_group_relin_fnn_step.argtypes = [_hp, ctypes.POINTER(almpc_opts)]
    L.almpc_group_relin_fnn_step_async.argtypes = [_hp, ctypes.POINTER(almpc_opts)]
    L.almpc_group_relin_fnn_advance.argtypes = [_hp]
    L.almpc_group_advance_plant.argtypes = [_hp]
    L.almpc_group_sqp_fnn_setup.argtypes = [_hp, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [_dp] * 10 + [ctypes.c_int, _dp, _dp, ctypes.c_double, ctypes.c_double]
    L.almpc_group_sqp_fnn_start.argtypes = [_hp, _dp, _dp]
    L.almpc_group_sqp_fnn_iterate.argtypes = [_hp, ctypes.c_int, ctypes.c_double, ctypes.POINTER(almpc_opts), _dp, _dp]
    L.almpc_group_sqp_fnn_skipped.argtypes = [_hp, _ip]
    L.almpc_group_x0_staging.argtypes = [_hp, ctypes.POINTER(_dp)]
    L.almpc_group_update_initialization_staged.argtypes = [_hp, ctypes.POINTER(_dp)]
    L.almpc_group_get_results_async.argtypes = [_hp, ctypes.c_uint32]
    L.almpc_group_get_results_wait.argtypes = [_hp, ctypes.c_int, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _ip]
    for nm_ in ("almpc_group_set_terminal_equality", "almpc_group_set_rho_profile", "almpc_group_set_structured_fallback", "almpc_group_set_state_box",
                "almpc_group_design_batched", "almpc_group_relin_fnn_setup", "almpc_group_relin_fnn_step", "almpc_group_relin_fnn_step_async",
                "almpc_group_relin_fnn_advance", "almpc_group_advance_plant", "almpc_group_sqp_fnn_set_structured", "almpc_group_sqp_fnn_set_step_rule",
                "almpc_group_sqp_fnn_setup", "almpc_group_sqp_fnn_start", "almpc_group_sqp_fnn_iterate", "almpc_group_sqp_fnn_skipped",
                "almpc_group_x0_staging", "almpc_group_update_initialization_staged", "almpc_group_get_results_async", "almpc_group_get_results_wait"):
        getattr(L, nm_).restype = ctypes.c_int
    L.almpc_set_start_from.argtypes = [_hp, _hp]
    L.almpc_set_start_from.restype = ctypes.c_int
    L.almpc_timing_samples.argtypes = [_hp, ctypes.c_int, ctypes.POINTER(ctypes.c_int), _fp, _fp, _fp, _fp]
    L.almpc_timing_samples.restype = ctypes.c_int
    for name in ("almpc_x0_staging", "almpc_update_initialization_async", "almpc_get_results_async", "almpc_get_results_wait", "almpc_host_results",
                 "almpc_get_first_input", "almpc_group_create", "almpc_group_size", "almpc_group_shard", "almpc_group_design_shared",
                 "almpc_group_set_reference", "almpc_group_update_initialization", "almpc_group_calculate",
                 "almpc_group_calculate_async", "almpc_group_synchronize", "almpc_group_get_results"):
        getattr(L, name).restype = ctypes.c_int
    for name in ("almpc_create", "almpc_design_shared", "almpc_set_reference", "almpc_update_initialization",
                 "almpc_update_initialization_device", "almpc_calculate", "almpc_calculate_async", "almpc_synchronize",
                 "almpc_get_results", "almpc_get_design", "almpc_device_results", "almpc_get_timing", "almpc_timing_reset",
                 "almpc_timing_summary"):
        getattr(L, name).restype = ctypes.c_int
    _lib = L
    return L


OPT_NO_WARM_STATE = 0x1  # almpc.h: ALMPC_OPT_NO_WARM_STATE (opts.reserved[0])
# almpc.h: ALMPC_WANT_* (almpc_get_results_async)
WANT = {"x": 0x01, "e_x": 0x02, "u": 0x04, "e_u": 0x08, "status": 0x10, "iters": 0x20, "polish_iters": 0x40, "u0": 0x80}


def _want_mask(want):
    try:
        return sum(WANT[k] for k in set(want))
    except KeyError as e:
        raise ValueError(f"unknown result {e.args[0]!r}; choose from {sorted(WANT)}") from None


def comm_unique_id() -> bytes:
    """128-byte RCCL id made by rank 0 (almpc_comm_unique_id); hand it to the other ranks by any channel."""
    buf = ctypes.create_string_buffer(128)
    rc = load().almpc_comm_unique_id(buf)
    if rc != ALMPC_OK:
        raise AlmpcError(rc, "almpc_comm_unique_id (librccl not loadable?)")
    return buf.raw


def default_opts(**kw) -> almpc_opts:
    """almpc_default_opts, then the given fields.  keep_warm_state=False sets ALMPC_OPT_NO_WARM_STATE (the step does not store the
    ADMM state a later warm start would need)."""
    o = almpc_opts()
    load().almpc_default_opts(ctypes.byref(o))
    if "keep_warm_state" in kw:
        if not kw.pop("keep_warm_state"):
            o.reserved[0] |= OPT_NO_WARM_STATE
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown solver option {k!r}")
        setattr(o, k, v)
    return o


def _colmajor(a, shape=None):
    a = np.asfortranarray(np.asarray(a, dtype=np.float64))
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def dare(A, B, Q, R):
    """P = DARE(A, B, Q, R) through the library (host code, as ControlSystems.are at src/sub/design_mpc.jl:327)."""
    L = load()
    A, B = np.asfortranarray(A, dtype=np.float64), np.asfortranarray(B, dtype=np.float64)
    n, m = B.shape
    Q, R = _colmajor(Q, (n, n)), _colmajor(R, (m, m))
    P = np.empty((n, n), order="F")
    rc = L.almpc_dare(n, m, _ptr(A), _ptr(B), _ptr(Q), _ptr(R), _ptr(P))
    if rc != ALMPC_OK:
        raise AlmpcError(rc, "almpc_dare")
    return P


FNN_ACTIVATIONS = {"identity": 0, "relu": 1, "tanh": 2, "sigmoid": 3, "swish": 4}


def fnn_linearize(W_in, W_h, b_h, W_out, x, u, act="relu", device=0, want_f=False):
    """Batched Jacobians of an Fnn model on the GPU: x (batch, n), u (batch, m) -> A (batch, n, n), B (batch, n, m)."""
    L = load()
    W_in, W_out = np.asfortranarray(W_in, dtype=np.float64), np.asfortranarray(W_out, dtype=np.float64)
    H, nin = W_in.shape
    n = W_out.shape[0]
    m = nin - n
    nl = len(W_h)
    Wh = np.ascontiguousarray(np.stack([np.asfortranarray(W, dtype=np.float64).T for W in W_h])) if nl else np.zeros((1, 1, 1))
    bh = np.ascontiguousarray(np.stack([np.asarray(b, dtype=np.float64) for b in b_h])) if nl else np.zeros((1, 1))
    x = np.ascontiguousarray(np.atleast_2d(x), dtype=np.float64)
    u = np.ascontiguousarray(np.atleast_2d(u), dtype=np.float64)
    batch = x.shape[0]
    A = np.empty((batch, n, n)); B = np.empty((batch, m, n)); f = np.empty((batch, n)) if want_f else None
    rc = L.almpc_fnn_linearize(int(device), n, m, H, nl, FNN_ACTIVATIONS[act], _ptr(W_in), _ptr(Wh), _ptr(bh), _ptr(W_out),
                               batch, _ptr(x), _ptr(u), _ptr(A), _ptr(B), _ptr(f))
    if rc != ALMPC_OK:
        raise AlmpcError(rc, "almpc_fnn_linearize")
    A, B = A.transpose(0, 2, 1), B.transpose(0, 2, 1)  # column-major buffers -> (batch, row, col)
    return (A, B, f) if want_f else (A, B)


class Solver:
    """Thin object wrapper of an almpc_handle: one device, one batch shard."""

    def __init__(self, n, m, N, batch, device=0, timing=False, structured=False, structured_fallback=None):
        """structured: ALMPC_FLAG_STRUCTURED (the stage-wise solve of the multiple-shooting form is the handle's solver: no m*N <= 128
        limit).  structured_fallback: the redo of the instances a condensed step leaves without a certificate by the stage-wise
        solvers -- None: the library default (on wherever they cover the design), True: required, False: off."""
        self.L = load()
        self.n, self.m, self.N, self.batch = int(n), int(m), int(N), int(batch)
        self.nz = self.m * self.N
        h = _hp()
        rc = self.L.almpc_create(ctypes.byref(h), self.n, self.m, self.N, self.batch, int(device),
                                 (FLAG_TIMING if timing else 0) | (FLAG_STRUCTURED if structured else 0))
        if rc != ALMPC_OK:
            raise AlmpcError(rc, "almpc_create failed (is a gfx950 GPU visible? there is no CPU fallback)")
        self.h = h
        if structured_fallback is not None:
            self._check(self.L.almpc_set_structured_fallback(self.h, 1 if structured_fallback else 0))

    def _check(self, rc):
        if rc != ALMPC_OK:
            raise AlmpcError(rc, (self.L.almpc_last_error(self.h) or b"").decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.almpc_destroy(self.h)
            self.h = None

    __del__ = close

    def design_shared(self, A, B, Q, R, S=None, P=None, umin=None, umax=None, xmin=None, xmax=None, rho=0.1, sigma=1e-6,
                      terminal="none", rho_profile="scalar"):
        """xmin/xmax: state box (the reference's kw mpc_state_constraint); terminal: "none" | "equality";
        rho_profile: "scalar" (OSQP: rho for every row) | "stiffness" (rho_i = rho / (H'^-1)_ii)."""
        n, m = self.n, self.m
        if terminal not in ("none", "equality"):
            raise ValueError("terminal must be 'none' or 'equality'")
        self._check(self.L.almpc_set_terminal_equality(self.h, 1 if terminal == "equality" else 0))
        self._check(self.L.almpc_set_rho_profile(self.h, {"scalar": 0, "stiffness": 1}[rho_profile]))
        A, B, Q, R = _colmajor(A, (n, n)), _colmajor(B, (n, m)), _colmajor(Q, (n, n)), _colmajor(R, (m, m))
        S = None if S is None else _colmajor(S, (m, m))
        P = None if P is None else _colmajor(P, (n, n))
        umin = np.ascontiguousarray(umin, dtype=np.float64).reshape(m)
        umax = np.ascontiguousarray(umax, dtype=np.float64).reshape(m)
        xmin = None if xmin is None else np.ascontiguousarray(xmin, dtype=np.float64).reshape(n)
        xmax = None if xmax is None else np.ascontiguousarray(xmax, dtype=np.float64).reshape(n)
        self._check(self.L.almpc_design_shared(self.h, _ptr(A), _ptr(B), _ptr(Q), _ptr(R), _ptr(S), _ptr(P), _ptr(umin),
                                               _ptr(umax), _ptr(xmin), _ptr(xmax), float(rho), float(sigma)))

    def _state_rows(self, xmin, xmax, terminal):
        """State box / terminal equality of the designs without xmin / xmax arguments (almpc_set_state_box, almpc_set_terminal_equality)."""
        if (xmin is None) != (xmax is None):
            raise ValueError("give both xmin and xmax or neither")
        self._check(self.L.almpc_set_terminal_equality(self.h, 1 if terminal == "equality" else 0))
        if xmin is None:
            self._check(self.L.almpc_set_state_box(self.h, None, None))
        else:
            lo = np.ascontiguousarray(xmin, dtype=np.float64).reshape(self.n)
            hi = np.ascontiguousarray(xmax, dtype=np.float64).reshape(self.n)
            self._check(self.L.almpc_set_state_box(self.h, _ptr(lo), _ptr(hi)))

    def design_batched(self, A_batch, B_batch, Q, R, S=None, P=None, umin=None, umax=None, rho=0.1, sigma=1e-6,
                       rho_profile="scalar", xmin=None, xmax=None, terminal="none"):
        """One model per instance: A_batch (batch, n, n), B_batch (batch, n, m).  P: None (DARE per instance), (n, n) shared
        or (batch, n, n).  xmin / xmax: state box (stages 1..N+1); terminal = "equality": e_x[:, N+1] = 0."""
        n, m, b = self.n, self.m, self.batch
        self._state_rows(xmin, xmax, terminal)
        self._check(self.L.almpc_set_rho_profile(self.h, {"scalar": 0, "stiffness": 1}[rho_profile]))
        A = np.ascontiguousarray(np.asarray(A_batch, dtype=np.float64).reshape(b, n, n).transpose(0, 2, 1))  # column-major blocks
        B = np.ascontiguousarray(np.asarray(B_batch, dtype=np.float64).reshape(b, n, m).transpose(0, 2, 1))
        Q, R = _colmajor(Q, (n, n)), _colmajor(R, (m, m))
        S = None if S is None else _colmajor(S, (m, m))
        p_inst = 0
        if P is not None:
            P = np.asarray(P, dtype=np.float64)
            if P.ndim == 3:
                P = np.ascontiguousarray(P.reshape(b, n, n).transpose(0, 2, 1)); p_inst = 1
            else:
                P = _colmajor(P, (n, n))
        umin = np.ascontiguousarray(umin, dtype=np.float64).reshape(m)
        umax = np.ascontiguousarray(umax, dtype=np.float64).reshape(m)
        self._check(self.L.almpc_design_batched(self.h, _ptr(A), _ptr(B), _ptr(Q), _ptr(R), _ptr(S), _ptr(P), p_inst, _ptr(umin),
                                                _ptr(umax), float(rho), float(sigma)))

    def design_ltv(self, A_all, B_all, c_all, xbar, ubar, x_ref, u_ref, Q, R, S=None, P=None, umin=None, umax=None, rho=0.1, sigma=1e-6,
                   rho_profile="scalar", xmin=None, xmax=None, terminal="none"):
        """Time-varying models: A_all (batch, N, n, n), B_all (batch, N, n, m), c_all (batch, N, n) or None, xbar (batch, n, N+1),
        ubar (batch, m, N), x_ref (n, N+1) / u_ref (m, N) or None, P (n, n) or (batch, n, n).  The QP variable is v = u - ubar.
        xmin / xmax: state box on xbar + dx (stages 1..N+1); terminal = "equality": xbar + dx = x_ref at stage N+1."""
        n, m, N, b = self.n, self.m, self.N, self.batch
        self._state_rows(xmin, xmax, terminal)
        self._check(self.L.almpc_set_rho_profile(self.h, {"scalar": 0, "stiffness": 1}[rho_profile]))
        A = np.ascontiguousarray(np.asarray(A_all, dtype=np.float64).reshape(b, N, n, n).transpose(0, 1, 3, 2))
        B = np.ascontiguousarray(np.asarray(B_all, dtype=np.float64).reshape(b, N, n, m).transpose(0, 1, 3, 2))
        c = None if c_all is None else np.ascontiguousarray(np.asarray(c_all, dtype=np.float64).reshape(b, N, n))
        xb = np.ascontiguousarray(np.asarray(xbar, dtype=np.float64).reshape(b, n, N + 1).transpose(0, 2, 1))
        ub = np.ascontiguousarray(np.asarray(ubar, dtype=np.float64).reshape(b, m, N).transpose(0, 2, 1))
        xr = None if x_ref is None else np.ascontiguousarray(np.asarray(x_ref, dtype=np.float64).reshape(n, N + 1).T)
        ur = None if u_ref is None else np.ascontiguousarray(np.asarray(u_ref, dtype=np.float64).reshape(m, N).T)
        Q, R = _colmajor(Q, (n, n)), _colmajor(R, (m, m))
        S = None if S is None else _colmajor(S, (m, m))
        P = np.asarray(P, dtype=np.float64)
        p_inst = 0
        if P.ndim == 3:
            P = np.ascontiguousarray(P.reshape(b, n, n).transpose(0, 2, 1)); p_inst = 1
        else:
            P = _colmajor(P, (n, n))
        umin = np.ascontiguousarray(umin, dtype=np.float64).reshape(m)
        umax = np.ascontiguousarray(umax, dtype=np.float64).reshape(m)
        self._check(self.L.almpc_design_ltv(self.h, _ptr(A), _ptr(B), _ptr(c), _ptr(xb), _ptr(ub), _ptr(xr), _ptr(ur), _ptr(Q), _ptr(R),
                                            _ptr(S), _ptr(P), p_inst, _ptr(umin), _ptr(umax), float(rho), float(sigma)))

    def sqp_fnn_setup(self, W_in, W_h, b_h, W_out, x_ref, u_ref, Q, R, S=None, P=None, umin=None, umax=None, act="relu", rho=0.1,
                      sigma=1e-6, rho_profile="scalar", xmin=None, xmax=None, terminal="none", qp_solver="condensed"):
        """SQP outer loop for an Fnn model (almpc_sqp_fnn_*): network as in fnn_linearize, x_ref (n, N+1) / u_ref (m, N) or None,
        P (n, n) or (batch, n, n).  qp_solver: "condensed" (default) or "structured" (every QP through k_riccati, no condensed design).  xmin / xmax: the state box of the reference's NLP branch
        (.../fnn/mpc_modeler_implementation_fnn.jl:146-153) as rows of every iteration's QP; terminal = "equality"."""
        n, m, N, b = self.n, self.m, self.N, self.batch
        self._state_rows(xmin, xmax, terminal)
        self._check(self.L.almpc_sqp_fnn_set_structured(self.h, 1 if qp_solver == "structured" else 0))
        self._check(self.L.almpc_set_rho_profile(self.h, {"scalar": 0, "stiffness": 1}[rho_profile]))
        W_in, W_out = np.asfortranarray(W_in, dtype=np.float64), np.asfortranarray(W_out, dtype=np.float64)
        H = W_in.shape[0]
        nl = len(W_h)
        Wh = np.ascontiguousarray(np.stack([np.asfortranarray(W, dtype=np.float64).T for W in W_h])) if nl else np.zeros((1, 1, 1))
        bh = np.ascontiguousarray(np.stack([np.asarray(v, dtype=np.float64) for v in b_h])) if nl else np.zeros((1, 1))
        xr = None if x_ref is None else np.ascontiguousarray(np.asarray(x_ref, dtype=np.float64).reshape(n, N + 1).T)
        ur = None if u_ref is None else np.ascontiguousarray(np.asarray(u_ref, dtype=np.float64).reshape(m, N).T)
        Q, R = _colmajor(Q, (n, n)), _colmajor(R, (m, m))
        S = None if S is None else _colmajor(S, (m, m))
        P = np.asarray(P, dtype=np.float64)
        p_inst = 0
        if P.ndim == 3:
            P = np.ascontiguousarray(P.reshape(b, n, n).transpose(0, 2, 1)); p_inst = 1
        else:
            P = _colmajor(P, (n, n))
        umin = np.ascontiguousarray(umin, dtype=np.float64).reshape(m)
        umax = np.ascontiguousarray(umax, dtype=np.float64).reshape(m)
        self._check(self.L.almpc_sqp_fnn_setup(self.h, H, nl, FNN_ACTIVATIONS[act], _ptr(W_in), _ptr(Wh), _ptr(bh), _ptr(W_out), _ptr(xr),
                                               _ptr(ur), _ptr(Q), _ptr(R), _ptr(S), _ptr(P), p_inst, _ptr(umin), _ptr(umax), float(rho),
                                               float(sigma)))

    def relin_fnn_setup(self, W_in, W_h, b_h, W_out, x_ref, u_ref, Q, R, S=None, P=None, umin=None, umax=None, act="relu", rho=0.1,
                        sigma=1e-6, rho_profile="scalar", xmin=None, xmax=None, terminal="none"):
        """Device-resident per-step re-linearisation of an Fnn model (almpc_relin_fnn_*, BASELINE configs[3]): network as in
        fnn_linearize, shared x_ref (n, N+1) / u_ref (m, N) or None, shared P (n, n); xmin / xmax / terminal as in design_batched."""
        n, m, N = self.n, self.m, self.N
        self._state_rows(xmin, xmax, terminal)
        self._check(self.L.almpc_set_rho_profile(self.h, {"scalar": 0, "stiffness": 1}[rho_profile]))
        W_in, W_out = np.asfortranarray(W_in, dtype=np.float64), np.asfortranarray(W_out, dtype=np.float64)
        H = W_in.shape[0]
        nl = len(W_h)
        Wh = np.ascontiguousarray(np.stack([np.asfortranarray(W, dtype=np.float64).T for W in W_h])) if nl else np.zeros((1, 1, 1))
        bh = np.ascontiguousarray(np.stack([np.asarray(v, dtype=np.float64) for v in b_h])) if nl else np.zeros((1, 1))
        xr = None if x_ref is None else np.ascontiguousarray(np.asarray(x_ref, dtype=np.float64).reshape(n, N + 1).T)
        ur = None if u_ref is None else np.ascontiguousarray(np.asarray(u_ref, dtype=np.float64).reshape(m, N).T)
        Q, R = _colmajor(Q, (n, n)), _colmajor(R, (m, m))
        S = None if S is None else _colmajor(S, (m, m))
        P = _colmajor(P, (n, n))
        umin = np.ascontiguousarray(umin, dtype=np.float64).reshape(m)
        umax = np.ascontiguousarray(umax, dtype=np.float64).reshape(m)
        self._check(self.L.almpc_relin_fnn_setup(self.h, H, nl, FNN_ACTIVATIONS[act], _ptr(W_in), _ptr(Wh), _ptr(bh), _ptr(W_out), _ptr(xr),
                                                 _ptr(ur), _ptr(Q), _ptr(R), _ptr(S), _ptr(P), _ptr(umin), _ptr(umax), float(rho),
                                                 float(sigma)))

    def relin_fnn_step(self, opts: almpc_opts | None = None, sync=True):
        fn = self.L.almpc_relin_fnn_step if sync else self.L.almpc_relin_fnn_step_async
        self._check(fn(self.h, ctypes.byref(opts) if opts is not None else None))

    def relin_fnn_advance(self):
        """x0 <- fnn(x0, u[:,1]) on the device (closed loop of the black-box model)."""
        self._check(self.L.almpc_relin_fnn_advance(self.h))

    def relin_fnn_timing(self):
        a, d, s = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
        self._check(self.L.almpc_relin_fnn_timing(self.h, ctypes.byref(a), ctypes.byref(d), ctypes.byref(s)))
        return dict(jacobian_ms=a.value, design_ms=d.value, step_ms=s.value)

    # ---- multi-GPU (RCCL inside the library): one process per GPU, one handle per process
    def comm_init(self, unique_id: bytes, rank: int, world: int):
        if len(unique_id) != 128:
            raise ValueError("the RCCL unique id is 128 bytes (comm_unique_id() on rank 0)")
        self._check(self.L.almpc_comm_init(self.h, unique_id, int(rank), int(world)))
        self._comm_world = int(world)

    def comm_summary(self):
        out = (ctypes.c_int64 * 4)()
        self._check(self.L.almpc_comm_summary(self.h, out))
        return dict(ranks=int(out[0]), unsolved=int(out[1]), admm_iters_max=int(out[2]), polish_iters_max=int(out[3]))

    def comm_allgather_first_input(self):
        """u[:, 1] of every instance of every rank: (world, batch, m)."""
        out = np.empty((self._comm_world, self.batch, self.m))
        self._check(self.L.almpc_comm_allgather_first_input(self.h, _ptr(out), None))
        return out

    def sqp_fnn_start(self, x0, u_guess=None):
        """x0 (batch, n); u_guess (batch, m, N) or None."""
        x0 = np.ascontiguousarray(np.asarray(x0, dtype=np.float64).reshape(self.batch, self.n))
        ug = None if u_guess is None else np.ascontiguousarray(np.asarray(u_guess, dtype=np.float64).reshape(self.batch, self.m, self.N).transpose(0, 2, 1))
        self._check(self.L.almpc_sqp_fnn_start(self.h, _ptr(x0), _ptr(ug)))

    def sqp_fnn_iterate(self, iters, step_scale=1.0, opts=None, step_rule="fixed"):
        """-> (step_inf[iters], defect_inf[iters]); raises AlmpcError(ALMPC_ERR_NUMERIC) when an instance had to skip an iteration
        (the histories are still filled: see .sqp_last).  step_rule: "fixed" | "merit" (l1 merit-function safeguard)."""
        self._check(self.L.almpc_sqp_fnn_set_step_rule(self.h, {"fixed": 0, "merit": 1}[step_rule]))
        st, de = np.zeros(int(iters)), np.zeros(int(iters))
        self.sqp_last = (st, de)
        self._check(self.L.almpc_sqp_fnn_iterate(self.h, int(iters), float(step_scale), ctypes.byref(opts) if opts is not None else None,
                                                 _ptr(st), _ptr(de)))
        return st, de

    def sqp_fnn_skipped(self):
        out = np.zeros(self.batch, dtype=np.int32)
        self._check(self.L.almpc_sqp_fnn_skipped(self.h, out.ctypes.data_as(_ip)))
        return out

    def get_gradient_instance(self, i):
        q = np.empty(self.nz)
        self._check(self.L.almpc_get_gradient_instance(self.h, int(i), _ptr(q)))
        return q

    def get_design_instance(self, i):
        n, nz = self.n, self.nz
        H = np.empty((nz, nz), order="F"); F = np.empty((nz, n), order="F"); d = np.empty(nz)
        self._check(self.L.almpc_get_design_instance(self.h, int(i), _ptr(H), _ptr(F), _ptr(d)))
        return dict(H=H, F=F, d=d)

    def start_from(self, other: "Solver"):
        """Structured handle: the next calculate starts from `other`'s last inputs (same batch, horizon <= this one's): horizon
        continuation / chaining of solvers (almpc_set_start_from)."""
        self._check(self.L.almpc_set_start_from(self.h, other.h))

    def set_step_fusion(self, on: bool):
        """One kernel per step (default) or the two-kernel path whose stage times can be told apart."""
        self._check(self.L.almpc_set_step_fusion(self.h, 1 if on else 0))

    def set_reference(self, x_ref, u_ref, per_instance=False):
        """x_ref (n, N+1) / u_ref (m, N), or with per_instance (batch, n, N+1) / (batch, m, N)."""
        n, m, N = self.n, self.m, self.N
        x_ref, u_ref = np.asarray(x_ref, dtype=np.float64), np.asarray(u_ref, dtype=np.float64)
        if per_instance:
            xr = np.ascontiguousarray(x_ref.reshape(self.batch, n, N + 1).transpose(0, 2, 1))
            ur = np.ascontiguousarray(u_ref.reshape(self.batch, m, N).transpose(0, 2, 1))
        else:
            xr = np.ascontiguousarray(x_ref.reshape(n, N + 1).T)
            ur = np.ascontiguousarray(u_ref.reshape(m, N).T)
        self._check(self.L.almpc_set_reference(self.h, _ptr(xr), _ptr(ur), 1 if per_instance else 0))

    def update_initialization(self, x0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(self.batch, self.n)
        self._check(self.L.almpc_update_initialization(self.h, _ptr(x0)))

    def update_initialization_device(self, dev_ptr: int):
        self._check(self.L.almpc_update_initialization_device(self.h, ctypes.c_void_p(dev_ptr)))

    def calculate(self, opts: almpc_opts | None = None, sync=True):
        fn = self.L.almpc_calculate if sync else self.L.almpc_calculate_async
        self._check(fn(self.h, None if opts is None else ctypes.byref(opts)))

    def synchronize(self):
        self._check(self.L.almpc_synchronize(self.h))

    def get_results(self, want=("x", "e_x", "u", "e_u", "status", "iters", "polish_iters")):
        b, n, m, N = self.batch, self.n, self.m, self.N
        bufs = {}
        if "x" in want: bufs["x"] = np.empty((b, N + 1, n))
        if "e_x" in want: bufs["e_x"] = np.empty((b, N + 1, n))
        if "u" in want: bufs["u"] = np.empty((b, N, m))
        if "e_u" in want: bufs["e_u"] = np.empty((b, N, m))
        for k in ("status", "iters", "polish_iters"):
            if k in want: bufs[k] = np.empty(b, dtype=np.int32)
        ip = lambda k: bufs[k].ctypes.data_as(_ip) if k in bufs else None
        self._check(self.L.almpc_get_results(self.h, _ptr(bufs.get("x")), _ptr(bufs.get("e_x")), _ptr(bufs.get("u")),
                                             _ptr(bufs.get("e_u")), ip("status"), ip("iters"), ip("polish_iters")))
        # Julia-shaped views: (batch, n, N+1) and (batch, m, N)
        for k in ("x", "e_x", "u", "e_u"):
            if k in bufs: bufs[k] = bufs[k].transpose(0, 2, 1)
        return bufs

    # ---- host-facing step path: pinned staging, transfers on copy streams (include/almpc.h "Host-facing step path")
    def update_initialization_async(self, x0):
        """x0 (batch, n) -> pinned slot -> upload on the copy-in stream; the next calculate waits for it on the device."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(self.batch, self.n)
        self._check(self.L.almpc_update_initialization_async(self.h, _ptr(x0)))

    def x0_staging(self):
        """Zero-copy input (almpc_x0_staging): a (batch, n) view of the pinned slot the next update_initialization_async will use.
        Fill it in place and pass it to update_initialization_async: no staging copy."""
        ptr = _dp()
        self._check(self.L.almpc_x0_staging(self.h, ctypes.byref(ptr)))
        return np.ctypeslib.as_array(ptr, shape=(self.batch, self.n))

    def get_results_async(self, want=("u0", "status")) -> int:
        """Ask for results of the last enqueued step (names of WANT; "u0" = u[:, 1] of every instance); returns a ticket."""
        t = self.L.almpc_get_results_async(self.h, _want_mask(want))
        if t < 0:
            self._check(t)
        return t

    def get_results_wait(self, ticket, want=("u0", "status"), copy=True):
        """Results of a ticket.  copy=True: fresh arrays (almpc_get_results_wait); copy=False: views of the handle's pinned slot
        (almpc_host_results: zero-copy, valid until two more requests).  Shapes as get_results; u0 (batch, m)."""
        b, n, m, N = self.batch, self.n, self.m, self.N
        shapes = {"x": (b, N + 1, n), "e_x": (b, N + 1, n), "u": (b, N, m), "e_u": (b, N, m), "u0": (b, m),
                  "status": (b,), "iters": (b,), "polish_iters": (b,)}
        order = ("x", "e_x", "u", "e_u", "u0", "status", "iters", "polish_iters")
        want = set(want)
        _want_mask(want)
        if copy:
            bufs = {k: np.empty(shapes[k], dtype=np.int32 if k in ("status", "iters", "polish_iters") else np.float64) for k in want}
            args = [(bufs[k].ctypes.data_as(_ip if bufs[k].dtype == np.int32 else _dp) if k in bufs else None) for k in order]
            self._check(self.L.almpc_get_results_wait(self.h, int(ticket), *args))
        else:
            self._check(self.L.almpc_get_results_wait(self.h, int(ticket), *([None] * 8)))
            ps = [ctypes.c_void_p() for _ in order]
            self._check(self.L.almpc_host_results(self.h, int(ticket), *[ctypes.byref(q) for q in ps]))
            bufs = {}
            for k, q in zip(order, ps):
                if k in want:
                    if not q.value:
                        raise AlmpcError(-1, f"result {k!r} was not part of the request of ticket {ticket}")
                    ct = ctypes.c_int32 if k in ("status", "iters", "polish_iters") else ctypes.c_double
                    cnt = int(np.prod(shapes[k]))
                    bufs[k] = np.ctypeslib.as_array(ctypes.cast(q.value, ctypes.POINTER(ct)), shape=(cnt,)).reshape(shapes[k])
        for k in ("x", "e_x", "u", "e_u"):
            if k in bufs: bufs[k] = bufs[k].transpose(0, 2, 1)
        return bufs

    def get_first_input(self):
        """u[:, 1] of every instance, (batch, m): what a receding-horizon caller applies (almpc_get_first_input)."""
        u0 = np.empty((self.batch, self.m))
        self._check(self.L.almpc_get_first_input(self.h, _ptr(u0)))
        return u0

    def get_design(self):
        n, nz = self.n, self.nz
        H = np.empty((nz, nz), order="F"); F = np.empty((nz, n), order="F"); P = np.empty((n, n), order="F"); d = np.empty(nz)
        self._check(self.L.almpc_get_design(self.h, _ptr(H), _ptr(F), _ptr(P), _ptr(d)))
        return dict(H=H, F=F, P=P, d=d)

    def device_results(self):
        ps = [ctypes.c_void_p() for _ in range(4)]
        self._check(self.L.almpc_device_results(self.h, *[ctypes.byref(p) for p in ps]))
        return dict(zip(("x", "e_x", "u", "e_u"), [p.value for p in ps]))

    def get_timing(self):
        v = [ctypes.c_float() for _ in range(4)]
        self._check(self.L.almpc_get_timing(self.h, *[ctypes.byref(x) for x in v]))
        return dict(zip(("admm_ms", "polish_ms", "rollout_ms", "total_ms"), [x.value for x in v]))

    def advance_plant(self):
        self._check(self.L.almpc_advance_plant(self.h))

    def debug_poison_lds(self):
        self._check(self.L.almpc_debug_poison_lds(self.h))

    def timing_set_stride(self, every):
        self._check(self.L.almpc_timing_set_stride(self.h, int(every)))

    def timing_reset(self, reserve_steps=0):
        self._check(self.L.almpc_timing_reset(self.h, int(reserve_steps)))

    def timing_samples(self, cap=4096):
        """Per recorded step its stage times (ms): dict of float32 arrays of length min(cap, steps recorded since the reset)."""
        n = ctypes.c_int()
        a = {k: np.zeros(int(cap), dtype=np.float32) for k in ("admm_ms", "polish_ms", "rollout_ms", "total_ms")}
        self._check(self.L.almpc_timing_samples(self.h, int(cap), ctypes.byref(n), *[a[k].ctypes.data_as(_fp) for k in a]))
        k = min(int(cap), n.value)
        return {key: v[:k] for key, v in a.items()}

    def timing_summary(self):
        n = ctypes.c_int()
        v = [ctypes.c_double() for _ in range(4)]
        self._check(self.L.almpc_timing_summary(self.h, ctypes.byref(n), *[ctypes.byref(x) for x in v]))
        out = dict(zip(("admm_ms", "polish_ms", "rollout_ms", "total_ms"), [x.value for x in v]))
        out["steps"] = n.value
        return out


class _HandleView(Solver):
    """A Solver over a handle owned by a Group (no create / destroy of its own)."""

    def __init__(self, L, h, n, m, N, batch):
        self.L, self.h = L, h
        self.n, self.m, self.N, self.batch = n, m, N, batch
        self.nz = m * N

    def close(self):
        self.h = None

    __del__ = close


class Group:
    """One process, several GPUs (almpc_group_*): one handle per entry of `devices` on its contiguous shard of the batch; every call
    fans out over the handles, nothing on the step path synchronises across devices.  Host arrays cover the whole batch."""

    def __init__(self, n, m, N, batch, devices, timing=False, structured=False, structured_fallback=None):
        self.L = load()
        self.n, self.m, self.N, self.batch = int(n), int(m), int(N), int(batch)
        self.nz = self.m * self.N
        devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        g = _hp()
        rc = self.L.almpc_group_create(ctypes.byref(g), self.n, self.m, self.N, self.batch, len(devices), devs,
                                       (FLAG_TIMING if timing else 0) | (FLAG_STRUCTURED if structured else 0))
        if rc != ALMPC_OK:
            raise AlmpcError(rc, "almpc_group_create failed (are the devices visible? there is no CPU fallback)")
        self.g = g
        self.shards = []
        for i in range(len(devices)):
            f, c = ctypes.c_int(), ctypes.c_int()
            self._check(self.L.almpc_group_shard(self.g, i, ctypes.byref(f), ctypes.byref(c)))
            self.shards.append((f.value, c.value))
        self.handles = [_HandleView(self.L, _hp(self.L.almpc_group_handle(self.g, i)), self.n, self.m, self.N, c)
                        for i, (_, c) in enumerate(self.shards)]
        if structured_fallback is not None:
            self._check(self.L.almpc_group_set_structured_fallback(self.g, 1 if structured_fallback else 0))

    def _check(self, rc):
        if rc != ALMPC_OK:
            raise AlmpcError(rc, (self.L.almpc_group_last_error(self.g) or b"").decode())

    def _state_rows(self, xmin, xmax, terminal):
        if (xmin is None) != (xmax is None):
            raise ValueError("give both xmin and xmax or neither")
        self._check(self.L.almpc_group_set_terminal_equality(self.g, 1 if terminal == "equality" else 0))
        lo = None if xmin is None else np.ascontiguousarray(xmin, dtype=np.float64).reshape(self.n)
        hi = None if xmax is None else np.ascontiguousarray(xmax, dtype=np.float64).reshape(self.n)
        self._check(self.L.almpc_group_set_state_box(self.g, _ptr(lo), _ptr(hi)))

    def design_batched(self, A_batch, B_batch, Q, R, S=None, P=None, umin=None, umax=None, rho=0.1, sigma=1e-6, rho_profile="scalar",
                       xmin=None, xmax=None, terminal="none"):
        """almpc_group_design_batched: arguments of Solver.design_batched for the whole batch."""
        n, m, b = self.n, self.m, self.batch
        self._state_rows(xmin, xmax, terminal)
        self._check(self.L.almpc_group_set_rho_profile(self.g, {"scalar": 0, "stiffness": 1}[rho_profile]))
        A = np.ascontiguousarray(np.asarray(A_batch, dtype=np.float64).reshape(b, n, n).transpose(0, 2, 1))
        B = np.ascontiguousarray(np.asarray(B_batch, dtype=np.float64).reshape(b, n, m).transpose(0, 2, 1))
        Q, R = _colmajor(Q, (n, n)), _colmajor(R, (m, m))
        S = None if S is None else _colmajor(S, (m, m))
        p_inst = 0
        if P is not None:
            P = np.asarray(P, dtype=np.float64)
            if P.ndim == 3:
                P = np.ascontiguousarray(P.reshape(b, n, n).transpose(0, 2, 1)); p_inst = 1
            else:
                P = _colmajor(P, (n, n))
        umin = np.ascontiguousarray(umin, dtype=np.float64).reshape(m)
        umax = np.ascontiguousarray(umax, dtype=np.float64).reshape(m)
        self._check(self.L.almpc_group_design_batched(self.g, _ptr(A), _ptr(B), _ptr(Q), _ptr(R), _ptr(S), _ptr(P), p_inst, _ptr(umin),
                                                      _ptr(umax), float(rho), float(sigma)))

    def _fnn_args(self, W_in, W_h, b_h, W_out, x_ref, u_ref, Q, R, S, P, umin, umax, act, p_batched):
        n, m, N = self.n, self.m, self.N
        W_in = np.asarray(W_in, dtype=np.float64)
        H = W_in.shape[0]
        W_h = [np.asarray(w, dtype=np.float64) for w in W_h]
        nl = len(W_h)
        Wh = np.ascontiguousarray(np.stack([w.T for w in W_h])) if nl else None
        bh = np.ascontiguousarray(np.stack([np.asarray(v, dtype=np.float64) for v in b_h])) if nl else None
        xr = np.ascontiguousarray(np.asarray(x_ref, dtype=np.float64).reshape(n, N + 1).T)
        ur = np.ascontiguousarray(np.asarray(u_ref, dtype=np.float64).reshape(m, N).T)
        P = np.asarray(P, dtype=np.float64)
        p_inst = 0
        if p_batched and P.ndim == 3:
            P = np.ascontiguousarray(P.reshape(self.batch, n, n).transpose(0, 2, 1)); p_inst = 1
        else:
            P = _colmajor(P, (n, n))
        arrs = [_colmajor(W_in, (H, n + m)), Wh, bh, _colmajor(W_out, (n, H)), xr, ur, _colmajor(Q, (n, n)), _colmajor(R, (m, m)),
                None if S is None else _colmajor(S, (m, m)), P,
                np.ascontiguousarray(umin, dtype=np.float64).reshape(m), np.ascontiguousarray(umax, dtype=np.float64).reshape(m)]
        return H, nl, FNN_ACTIVATIONS[act], arrs, p_inst

    def relin_fnn_setup(self, W_in, W_h, b_h, W_out, x_ref, u_ref, Q, R, S=None, P=None, umin=None, umax=None, act="relu", rho=0.1,
                        sigma=1e-6, rho_profile="scalar", xmin=None, xmax=None, terminal="none"):
        self._state_rows(xmin, xmax, terminal)
        self._check(self.L.almpc_group_set_rho_profile(self.g, {"scalar": 0, "stiffness": 1}[rho_profile]))
        H, nl, a, arrs, _ = self._fnn_args(W_in, W_h, b_h, W_out, x_ref, u_ref, Q, R, S, P, umin, umax, act, False)
        self._keep = arrs
        self._check(self.L.almpc_group_relin_fnn_setup(self.g, H, nl, a, *[_ptr(v) for v in arrs], float(rho), float(sigma)))

    def relin_fnn_step(self, opts: almpc_opts | None = None, sync=True):
        fn = self.L.almpc_group_relin_fnn_step if sync else self.L.almpc_group_relin_fnn_step_async
        self._check(fn(self.g, None if opts is None else ctypes.byref(opts)))

    def relin_fnn_advance(self):
        self._check(self.L.almpc_group_relin_fnn_advance(self.g))

    def advance_plant(self):
        self._check(self.L.almpc_group_advance_plant(self.g))

    def sqp_fnn_setup(self, W_in, W_h, b_h, W_out, x_ref, u_ref, Q, R, S=None, P=None, umin=None, umax=None, act="relu", rho=0.1,
                      sigma=1e-6, rho_profile="scalar", xmin=None, xmax=None, terminal="none", qp_solver="condensed"):
        self._state_rows(xmin, xmax, terminal)
        self._check(self.L.almpc_group_set_rho_profile(self.g, {"scalar": 0, "stiffness": 1}[rho_profile]))
        self._check(self.L.almpc_group_sqp_fnn_set_structured(self.g, 1 if qp_solver == "structured" else 0))
        H, nl, a, arrs, p_inst = self._fnn_args(W_in, W_h, b_h, W_out, x_ref, u_ref, Q, R, S, P, umin, umax, act, True)
        self._keep = arrs
        ptrs = [_ptr(v) for v in arrs]
        self._check(self.L.almpc_group_sqp_fnn_setup(self.g, H, nl, a, *ptrs[:10], p_inst, ptrs[10], ptrs[11], float(rho), float(sigma)))

    def sqp_fnn_start(self, x0, u_guess=None):
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(self.batch, self.n)
        ug = None if u_guess is None else np.ascontiguousarray(np.asarray(u_guess, dtype=np.float64).reshape(self.batch, self.m, self.N).transpose(0, 2, 1))
        self._check(self.L.almpc_group_sqp_fnn_start(self.g, _ptr(x0), _ptr(ug)))

    def sqp_fnn_iterate(self, iters, step_scale=1.0, opts=None, step_rule="fixed"):
        self._check(self.L.almpc_group_sqp_fnn_set_step_rule(self.g, {"fixed": 0, "merit": 1}[step_rule]))
        st, de = np.zeros(int(iters)), np.zeros(int(iters))
        self._check(self.L.almpc_group_sqp_fnn_iterate(self.g, int(iters), float(step_scale), None if opts is None else ctypes.byref(opts),
                                                       _ptr(st), _ptr(de)))
        return st, de

    def sqp_fnn_skipped(self):
        sk = np.zeros(self.batch, dtype=np.int32)
        self._check(self.L.almpc_group_sqp_fnn_skipped(self.g, sk.ctypes.data_as(_ip)))
        return sk

    def x0_staging(self):
        """The handles' pinned x0 slots as numpy views ((count_i, n) each): write the shards' states there, then update_initialization_staged()."""
        slots = (_dp * len(self.handles))()
        self._check(self.L.almpc_group_x0_staging(self.g, slots))
        self._slots = slots
        return [np.ctypeslib.as_array(slots[i], shape=(c, self.n)) for i, (_, c) in enumerate(self.shards)]

    def update_initialization_staged(self):
        self._check(self.L.almpc_group_update_initialization_staged(self.g, self._slots))

    def get_results_async(self, want=("u0", "status")) -> int:
        t = self.L.almpc_group_get_results_async(self.g, _want_mask(want))
        if t < 0:
            self._check(t)
        return t

    def get_results_wait(self, ticket, want=("u0", "status")):
        b, n, m, N = self.batch, self.n, self.m, self.N
        shapes = {"x": (b, N + 1, n), "e_x": (b, N + 1, n), "u": (b, N, m), "e_u": (b, N, m), "u0": (b, m),
                  "status": (b,), "iters": (b,), "polish_iters": (b,)}
        order = ("x", "e_x", "u", "e_u", "u0", "status", "iters", "polish_iters")
        bufs = {k: np.empty(shapes[k], dtype=np.int32 if k in ("status", "iters", "polish_iters") else np.float64) for k in set(want)}
        args = [(bufs[k].ctypes.data_as(_ip if bufs[k].dtype == np.int32 else _dp) if k in bufs else None) for k in order]
        self._check(self.L.almpc_group_get_results_wait(self.g, int(ticket), *args))
        for k in ("x", "e_x", "u", "e_u"):
            if k in bufs: bufs[k] = bufs[k].transpose(0, 2, 1)
        return bufs

    def close(self):
        if getattr(self, "g", None):
            for hv in self.handles:
                hv.close()
            self.L.almpc_group_destroy(self.g)
            self.g = None

    __del__ = close

    def design_shared(self, A, B, Q, R, S=None, P=None, umin=None, umax=None, xmin=None, xmax=None, rho=0.1, sigma=1e-6,
                      terminal="none", rho_profile="scalar"):
        n, m = self.n, self.m
        for hv in self.handles:   # per-handle options go through the handles (almpc_group_handle)
            hv._check(self.L.almpc_set_terminal_equality(hv.h, 1 if terminal == "equality" else 0))
            hv._check(self.L.almpc_set_rho_profile(hv.h, {"scalar": 0, "stiffness": 1}[rho_profile]))
        A, B, Q, R = _colmajor(A, (n, n)), _colmajor(B, (n, m)), _colmajor(Q, (n, n)), _colmajor(R, (m, m))
        S = None if S is None else _colmajor(S, (m, m))
        P = None if P is None else _colmajor(P, (n, n))
        umin = np.ascontiguousarray(umin, dtype=np.float64).reshape(m)
        umax = np.ascontiguousarray(umax, dtype=np.float64).reshape(m)
        xmin = None if xmin is None else np.ascontiguousarray(xmin, dtype=np.float64).reshape(n)
        xmax = None if xmax is None else np.ascontiguousarray(xmax, dtype=np.float64).reshape(n)
        self._check(self.L.almpc_group_design_shared(self.g, _ptr(A), _ptr(B), _ptr(Q), _ptr(R), _ptr(S), _ptr(P), _ptr(umin),
                                                     _ptr(umax), _ptr(xmin), _ptr(xmax), float(rho), float(sigma)))

    def set_reference(self, x_ref, u_ref, per_instance=False):
        n, m, N = self.n, self.m, self.N
        x_ref, u_ref = np.asarray(x_ref, dtype=np.float64), np.asarray(u_ref, dtype=np.float64)
        if per_instance:
            xr = np.ascontiguousarray(x_ref.reshape(self.batch, n, N + 1).transpose(0, 2, 1))
            ur = np.ascontiguousarray(u_ref.reshape(self.batch, m, N).transpose(0, 2, 1))
        else:
            xr = np.ascontiguousarray(x_ref.reshape(n, N + 1).T)
            ur = np.ascontiguousarray(u_ref.reshape(m, N).T)
        self._check(self.L.almpc_group_set_reference(self.g, _ptr(xr), _ptr(ur), 1 if per_instance else 0))

    def update_initialization(self, x0, resident=False):
        """x0 (batch, n) to the handles' shards.  Default: almpc_group_update_initialization (pinned slots read in place by the next
        step, no device copy -- the per-step path).  resident=True: a device copy per handle (almpc_update_initialization), for states
        that many steps will read (the pinned slot costs every step its transfer over the link)."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(self.batch, self.n)
        if resident:
            for hv, (f, c) in zip(self.handles, self.shards):
                hv.update_initialization(x0[f:f + c])
            return
        self._check(self.L.almpc_group_update_initialization(self.g, _ptr(x0)))

    def calculate(self, opts: almpc_opts | None = None, sync=True):
        fn = self.L.almpc_group_calculate if sync else self.L.almpc_group_calculate_async
        self._check(fn(self.g, None if opts is None else ctypes.byref(opts)))

    def synchronize(self):
        self._check(self.L.almpc_group_synchronize(self.g))

    def get_results(self, want=("x", "e_x", "u", "e_u", "status", "iters", "polish_iters")):
        b, n, m, N = self.batch, self.n, self.m, self.N
        shapes = {"x": (b, N + 1, n), "e_x": (b, N + 1, n), "u": (b, N, m), "e_u": (b, N, m), "u0": (b, m),
                  "status": (b,), "iters": (b,), "polish_iters": (b,)}
        order = ("x", "e_x", "u", "e_u", "u0", "status", "iters", "polish_iters")
        _want_mask(want)
        bufs = {k: np.empty(shapes[k], dtype=np.int32 if k in ("status", "iters", "polish_iters") else np.float64) for k in set(want)}
        args = [(bufs[k].ctypes.data_as(_ip if bufs[k].dtype == np.int32 else _dp) if k in bufs else None) for k in order]
        self._check(self.L.almpc_group_get_results(self.g, *args))
        for k in ("x", "e_x", "u", "e_u"):
            if k in bufs: bufs[k] = bufs[k].transpose(0, 2, 1)
        return bufs
