"""Debug aid: fused design chain against the split launches (ALMPC_DBG_SPLIT_*), instance by instance against the exact oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
n, m, N, b = 4, 2, 20, 200
f = mo.synthetic_fnn(act="relu")
x_ref = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N + 1)); u_ref = np.tile(np.array([0.1, -0.2])[:, None], (1, N))
X0 = x_ref[:, 0][None, :] + mo.splitmix_normal(0x5EED0004, 0, b, n)
A = np.empty((b, n, n)); B = np.empty((b, n, m))
for i in range(b):
    A[i], B[i] = f.jacobian(X0[i], u_ref[:, 0])
Q, R, P = 100.0 * np.eye(n), 0.1 * np.eye(m), 150.0 * np.eye(n)
sw = ("ALMPC_DBG_SPLIT_SCALE", "ALMPC_DBG_SPLIT_NEGGM", "ALMPC_DBG_SPLIT_INVERSES")
res = {}
for tag, on in (("fused", ()), ("split", sw), ("only_scale", sw[:1]), ("only_neggm", sw[1:2]), ("only_inv", sw[2:])):
    for k in sw:
        os.environ.pop(k, None)
    for k in on:
        os.environ[k] = "1"
    s = capi.Solver(n, m, N, b)
    s.design_batched(A, B, Q, R, None, P, -np.ones(m), np.ones(m), rho=0.1)
    s.set_reference(x_ref, u_ref); s.update_initialization(X0)
    s.calculate(capi.default_opts(rho=0.1))
    res[tag] = s.get_results()
    s.close()
for tag in res:
    d = np.abs(res[tag]["u"] - res["split"]["u"]).reshape(b, -1).max(axis=1)
    i = int(np.argmax(d))
    print(tag, "max |u - u_split|", d.max(), "instance", i, "iters", res[tag]["iters"][i], res["split"]["iters"][i], "polish", res[tag]["polish_iters"][i], res["split"]["polish_iters"][i])
i = int(np.argmax(np.abs(res["fused"]["u"] - res["split"]["u"]).reshape(b, -1).max(axis=1)))
pi = mo.make_problem(A[i], B[i], N, -np.ones(m), np.ones(m), x_ref=x_ref, u_ref=u_ref, P=P)
e = mo.solve_mpc_exact(pi, X0[i])["u"]
for tag in res:
    print(tag, "against the exact oracle, instance", i, np.abs(res[tag]["u"][i] - e).max())
