import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
p = mo.quadrotor()
X0 = mo.quadrotor_x0_batch(20, 1.0)
X0[3, 2] = np.nan
X0[18, 0] = np.inf
for structured in (True, False):
    for fb in ((None,) if structured else (0, 1)):
        s = capi.Solver(12, 4, 30, 20, structured=structured)
        if fb is not None: s._check(s.L.almpc_set_structured_fallback(s.h, fb))
        s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max)
        s.update_initialization(X0); s.calculate()
        r = s.get_results()
        print("structured", structured, "fallback", fb, "status", r["status"].tolist(), "piters", r["polish_iters"].tolist())
        s.close()
