"""Which instances of the structured state-row batches come back without a certificate, and what the restatement says about them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo, stagewise_oracle as so
capi = almpc_loader.load_package()._capi
b = 4096
X0 = np.concatenate([mo.quadrotor_x0_batch(b // 4 if a != 1.0 else b // 2, a, first_instance=k * b) for k, a in enumerate((0.3, 1.0, 3.0))])[:b]
XMAX = np.array([3, 3, 3, 1.5, 1.5, 1.5, 0.3, 0.3, 0.3, 1.0, 1.0, 1.0])
N = 50
q = mo.quadrotor(N)
for name, kw in (("state box", dict(x_min=-XMAX, x_max=XMAX)), ("box+eq", dict(x_min=-XMAX, x_max=XMAX, terminal="equality"))):
    p = mo.make_problem(q.A, q.B, N, q.u_min, q.u_max, **kw)
    X = np.clip(X0, -0.99 * XMAX, 0.99 * XMAX)
    s = capi.Solver(12, 4, N, b, structured=True)
    s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=p.x_min, xmax=p.x_max, terminal=p.terminal)
    s.set_reference(p.x_ref, p.u_ref); s.update_initialization(X)
    s.calculate()
    r = s.get_results(want=("status", "polish_iters"))
    bad = np.nonzero(r["status"] == 1)[0]
    print(name, "unsolved", len(bad), "piters", r["polish_iters"][bad][:20])
    for i in bad[:6]:
        o = so.solve_stage_dual(so.stage_qp_from_problem(p, X[i]))
        o64 = so.solve_stage_dual(so.stage_qp_from_problem(p, X[i]), wcap=64)
        print("  inst", i, "oracle status", o["status"], "iters", o["iters"], "n_active", o["n_active"], "| wcap 64:", o64["status"], o64["iters"], o64["n_active"],
              "slack", mo.feasibility_slack(p, X[i]))
    s.close()
