import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
import importlib.util
spec = importlib.util.spec_from_file_location("t", "/root/repo/tests/test_gpu_parity.py"); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
for name in ("di_eq","di_box_eq","quad_eq"):
    p, X0 = t._constrained_problems(mo)[name]
    s = capi.Solver(p.n, p.m, p.N, len(X0))
    s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=p.x_min, xmax=p.x_max, terminal=p.terminal)
    s.set_reference(p.x_ref, p.u_ref); s.update_initialization(X0); s.calculate(); r = s.get_results(); s.close()
    for i in range(min(len(X0),6)):
        try:
            e = mo.solve_mpc_exact(p, X0[i], return_info=True); print(name, i, "status", r["status"][i], "du %.2e"%np.abs(r["u"][i]-e["u"]).max(), "terminal %.2e"%np.abs(r["e_x"][i][:,-1]).max(), "pits", r["polish_iters"][i])
        except ValueError: print(name, i, "oracle infeasible; status", r["status"][i])
