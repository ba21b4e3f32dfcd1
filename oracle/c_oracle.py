"""ctypes binding of oracle/almpc_oracle.c (TEST INFRASTRUCTURE ONLY -- see that file's header).

Used by tests/ (C restatement vs numpy restatement vs HIP path) and by bench.py's `cpu_baseline`
leg (kind "port").  Build with `make oracle` (gcc -O3 -march=native -fopenmp)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def _cpu_tag():
    """-march=native code must not travel between hosts: key the build on the CPU's feature flags."""
    import hashlib
    try:
        with open("/proc/cpuinfo") as f:
            flags = next((ln for ln in f if ln.startswith("flags")), "")
    except OSError:
        flags = ""
    return hashlib.sha1(flags.encode()).hexdigest()[:10]


_SO = os.path.join(_HERE, "_build", f"libalmpc_oracle.{_cpu_tag()}.so")
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    src = os.path.join(_HERE, "almpc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", "-Wall",
                               "-o", _SO, src, "-lm"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.almpc_oracle_step_batch.restype = ctypes.c_int
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def step_batch(p, des, X0, alpha=1.6, eps_abs=1e-3, eps_rel=1e-3, max_iter=25, check_every=25, polish=True,
               polish_max_iter=0, threads=0):
    """One MPC step for every row of X0 ([batch][n]) with the C restatement.  `p` is an
    mpc_oracle.MPCProblem (shared references: column 0 of x_ref/u_ref is NOT assumed constant, the
    full horizon arrays are passed), `des` = mpc_oracle.design_shared(p, rho, sigma)."""
    n, m, N = p.n, p.m, p.N
    nz = m * N
    X0 = np.ascontiguousarray(X0, dtype=np.float64).reshape(-1, n)
    batch = X0.shape[0]
    f = lambda a: np.asfortranarray(a, dtype=np.float64)  # column-major
    A, B, Minv, Hs, G, Fs = f(p.A), f(p.B), f(des["Minv"]), f(des["Hs"]), f(des["G"]), f(des["Fs"])
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    fS, lo, hi, d = c(des["fS"]), c(des["lo"]), c(des["hi"]), c(des["d"])
    rho_vec = c(des["rho_vec"])
    xref, uref = f(p.x_ref), f(p.u_ref)  # column-major n x (N+1) == [N+1][n]
    x = np.empty((batch, N + 1, n))
    e_x = np.empty((batch, N + 1, n))
    u = np.empty((batch, N, m))
    e_u = np.empty((batch, N, m))
    status = np.empty(batch, dtype=np.int32)
    iters = np.empty(batch, dtype=np.int32)
    piters = np.empty(batch, dtype=np.int32)
    used = lib().almpc_oracle_step_batch(
        n, m, N, batch, _d(A), _d(B), _d(Minv), _d(Hs), _d(G), _d(Fs), _d(fS), _d(lo), _d(hi), _d(d), _d(xref),
        _d(uref), _d(X0), _d(rho_vec), ctypes.c_double(des["sigma"]), ctypes.c_double(alpha),
        ctypes.c_double(eps_abs), ctypes.c_double(eps_rel), int(max_iter), int(check_every), int(bool(polish)),
        int(polish_max_iter), _d(x), _d(e_x), _d(u), _d(e_u), status.ctypes.data_as(_ip), iters.ctypes.data_as(_ip),
        piters.ctypes.data_as(_ip), int(threads))
    if used < 0:
        raise MemoryError("almpc_oracle_step_batch: allocation failed")
    # Julia-shaped views: x (batch, n, N+1), u (batch, m, N)
    return dict(x=x.transpose(0, 2, 1), e_x=e_x.transpose(0, 2, 1), u=u.transpose(0, 2, 1), e_u=e_u.transpose(0, 2, 1),
                status=status, iters=iters, polish_iters=piters, threads=used)
