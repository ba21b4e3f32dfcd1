import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo, c_oracle as co
capi = almpc_loader.load_package()._capi
p = mo.quadrotor()
X0 = np.concatenate([mo.quadrotor_x0_batch(8, a, first_instance=160 * k) for k, a in enumerate((0.3, 1.0, 3.0, 6.0))])
des = mo.design_shared(p, rho=30.0, rho_profile="stiffness")
s = capi.Solver(p.n, p.m, p.N, len(X0))
s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, rho=30.0, rho_profile="stiffness")
s.set_reference(p.x_ref, p.u_ref); s.update_initialization(X0)
for K in (1, 2, 8):
    s.calculate(capi.default_opts(rho=30.0, max_iter=K, check_every=K, polish=0)); r = s.get_results()
    c = co.step_batch(p, des, X0, max_iter=K, check_every=K, polish=False)
    a = np.array([mo.admm_box(des["Hs"], des["Fs"] @ X0[i], des["lo"], des["hi"], rho=des["rho_vec"], Minv=des["Minv"], unscale=des["d"], max_iter=K, check_every=K)["z"] * des["d"] for i in range(len(X0))])
    eg = r["e_u"].transpose(0, 2, 1).reshape(len(X0), -1); ec = c["e_u"].transpose(0, 2, 1).reshape(len(X0), -1)
    print("K", K, "gpu vs C per instance:", np.abs(eg - ec).max(1).round(12))
    print("      C vs numpy:", np.abs(ec - a).max(), " gpu vs numpy:", np.abs(eg - a).max())
