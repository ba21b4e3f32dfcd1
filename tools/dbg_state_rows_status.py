"""Debug aid: full-batch quadrotor with tight box + terminal equality; saves the per-instance status / iterations to gpurun_out/."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
q = mo.quadrotor(); batch, N = 4096, 30
xmax = np.array([1, 1, 1, .5, .5, .5, .1, .1, .1, .1, .1, .1]) * 3.0
X0 = np.clip(mo.quadrotor_x0_batch(batch, 1.0), -0.99 * xmax, 0.99 * xmax)
s = capi.Solver(12, 4, N, batch)
s.design_shared(q.A, q.B, q.Q, q.R, q.S, None, q.u_min, q.u_max, xmin=-xmax, xmax=xmax, terminal="equality", rho=30.0, rho_profile="stiffness")
s.update_initialization(X0)
s.calculate(capi.default_opts(rho=30.0, max_iter=8, check_every=8))
r = s.get_results()
print(np.bincount(r["status"], minlength=4), "uncertified:", np.flatnonzero(r["status"] == 1).tolist())
os.makedirs(os.path.join(ROOT, "gpurun_out", "r3"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "r3", "state_rows_status.npz"), status=r["status"], piters=r["polish_iters"], u=r["u"][r["status"] == 1])
s.close()
