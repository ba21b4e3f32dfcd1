"""Profiling helper: two batched designs of 4096 quadrotor-size models (for rocprofv3 --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
q = mo.quadrotor(); b = 4096
A = np.repeat(q.A[None], b, 0); B = q.B[None] * (1.0 + 0.05 * np.sin(np.arange(b)))[:, None, None]
s = capi.Solver(12, 4, 30, b)
for _ in range(2): s.design_batched(A, B, q.Q, q.R, None, q.P, q.u_min, q.u_max)
s.close()
