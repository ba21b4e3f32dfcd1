import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
f = mo.synthetic_fnn()
b = 1024; N = 20
x_ref, u_ref = np.array([0.2, -0.1, 0.05, 0.0]), np.array([0.1, -0.2])
X0 = x_ref[None, :] + mo.splitmix_normal(0x5EED0004, 0, b, 4) * 2.0
A, B = capi.fnn_linearize(f.W_in, f.W_h, f.b_h, f.W_out, X0, np.repeat(u_ref[None], b, 0), act=f.act)
for P, nm in ((None, "dare"), (100.0 * np.eye(4), "100I")):
    s = capi.Solver(4, 2, N, b)
    s.design_batched(A, B, 100 * np.eye(4), 0.1 * np.eye(2), None, P, [-1, -1], [1, 1], rho=30.0, rho_profile="stiffness")
    s.update_initialization(X0)
    s.calculate(capi.default_opts(rho=30.0, max_iter=8, check_every=8))
    r = s.get_results()
    pit = r["polish_iters"]
    print(nm, "status", np.bincount(r["status"]).tolist(), "polish its max", pit.max(), "top", np.argsort(-pit)[:8].tolist(), np.sort(pit)[-8:].tolist())
    i = int(np.argmax(pit))
    p = mo.make_problem(A[i], B[i], N, [-1, -1], [1, 1], P=P)
    des = mo.design_shared(p, rho=30.0, rho_profile="stiffness")
    fs = des["Fs"] @ (X0[i] - p.x_ref[:, 0]) + des["fS"]
    a = mo.admm_box(des["Hs"], fs, des["lo"], des["hi"], rho=des["rho_vec"], sigma=des["sigma"], max_iter=8, check_every=8, Minv=des["Minv"], unscale=des["d"])
    pol = mo.polish_active_set(des["G"], -des["G"] @ fs, des["lo"], des["hi"], a["z"], a["y"])
    print("  instance", i, "oracle polish iters", pol["iters"], "adds", pol["n_add"], "removes", pol["n_remove"], "cond(Hs)", np.linalg.cond(des["Hs"]), "eig(A)", np.abs(np.linalg.eigvals(A[i])).round(3))
    e = mo.solve_mpc_exact(p, X0[i])
    print("  u err vs exact", np.abs(r["u"][i] - e["u"]).max(), "active", int(((e["u"] <= -1) | (e["u"] >= 1)).sum()))
    s.close()
