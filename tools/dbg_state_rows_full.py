"""Debug aid: full-batch quadrotor with tight box + terminal equality; list solved instances that violate the box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
q = mo.quadrotor(); batch, N = 4096, 30
xmax = np.array([1, 1, 1, .5, .5, .5, .1, .1, .1, .1, .1, .1]) * 3.0
p = mo.make_problem(q.A, q.B, N, q.u_min, q.u_max, x_min=-xmax, x_max=xmax, terminal="equality")
X0 = np.clip(mo.quadrotor_x0_batch(batch, 1.0), -0.99 * xmax, 0.99 * xmax)
s = capi.Solver(12, 4, N, batch)
s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=-xmax, xmax=xmax, terminal="equality", rho=30.0, rho_profile="stiffness")
s.update_initialization(X0)
s.calculate(capi.default_opts(rho=30.0, max_iter=8, check_every=8))
r = s.get_results()
st = r["status"]; print(np.bincount(st, minlength=4))
v = (np.abs(r["x"]) - xmax[None, :, None]).reshape(batch, -1).max(axis=1)
bad = np.flatnonzero((st == 0) & (v > 1e-6))
print("solved but violating:", len(bad), bad[:20], "polish its", r["polish_iters"][bad[:20]])
for i in bad[:5]:
    vi = np.abs(r["x"][i]) - xmax[:, None]
    si, ki = np.unravel_index(np.argmax(vi), vi.shape)
    try:
        e = mo.solve_mpc_exact(p, X0[i], return_info=True)
    except RuntimeError as ex:
        print(i, "viol %.3g" % vi.max(), "oracle does not certify:", str(ex)[-60:]); continue
    try:
        pass
        print(i, "viol %.3g at state %d stage %d" % (vi.max(), si, ki), "oracle: |du| %.2e active state rows %d total W %d" % (np.abs(r["u"][i] - e["u"]).max(), e["info"]["n_active_state"], len(e["info"]["W"])))
    except ValueError as ex:
        print(i, "viol %.3g" % vi.max(), "oracle: infeasible")
s.close()
