"""Diagnostic: which instances of the relinearised Fnn batch (tests/test_gpu_batched_models.py::test_config4...) hit the active-set cap."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
f = mo.synthetic_fnn(); batch, N = 1024, 20
x_ref, u_ref = np.array([0.2, -0.1, 0.05, 0.0]), np.array([0.1, -0.2])
X0 = x_ref[None, :] + mo.splitmix_normal(0x5EED0004, 0, batch, 4) * 2.0
A, B = capi.fnn_linearize(f.W_in, f.W_h, f.b_h, f.W_out, X0, np.repeat(u_ref[None], batch, 0), act=f.act)
s = capi.Solver(4, 2, N, batch)
s.design_batched(A, B, 100.0 * np.eye(4), 0.1 * np.eye(2), None, None, [-1, -1], [1, 1])
s.set_reference(x_ref[:, None] * np.ones((4, N + 1)), u_ref[:, None] * np.ones((2, N)))
s.update_initialization(X0)
s.calculate()
r = s.get_results()
bad = np.nonzero(r["status"] != 0)[0]
print("unsolved:", bad, r["status"][bad], "piters", r["polish_iters"][bad], "max piters overall", r["polish_iters"].max(), "p99", np.percentile(r["polish_iters"], 99))
for i in bad:
    d = s.get_design_instance(int(i))
    Hs = d["H"] * np.outer(d["d"], d["d"])
    print(i, "cond(H')", np.linalg.cond(Hs), "rho(A)", np.max(np.abs(np.linalg.eigvals(A[i]))), "min eig", np.linalg.eigvalsh(Hs)[:3])
    p = mo.make_problem(A[i], B[i], N, [-1, -1], [1, 1], x_ref=x_ref, u_ref=u_ref)
    e = mo.solve_mpc_exact(p, X0[i])
    print("   exact active", int(((e["u"] <= -1) | (e["u"] >= 1)).sum()), "u err", np.abs(r["u"][i] - e["u"]).max())
    np.savez(os.path.join(ROOT, "gpurun_out", f"cyc_{i}.npz"), A=A[i], B=B[i], x0=X0[i])
s.close()
