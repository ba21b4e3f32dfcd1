"""Debug aid / timing: SQP loop with the condensed and with the structured QP solver (256 instances, N = 50)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
f = mo.synthetic_fnn(act="tanh"); n, m, N, b = 4, 2, 50, 256
x_ref = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N + 1)); u_ref = np.tile(np.array([0.1, -0.2])[:, None], (1, N))
X0 = x_ref[:, 0][None, :] + 0.6 * mo.splitmix_normal(0x5EED0005, 0, b, n)
Q, R, P = 100.0 * np.eye(n), 0.1 * np.eye(m), 150.0 * np.eye(n)
res = {}
for mode in ("condensed", "structured"):
    s = capi.Solver(n, m, N, b)
    s.sqp_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, None, P, -np.ones(m), np.ones(m), act="tanh", qp_solver=mode)
    s.sqp_fnn_start(X0); s.sqp_fnn_iterate(5)
    s.sqp_fnn_start(X0)
    t0 = time.perf_counter(); st, de = s.sqp_fnn_iterate(20, step_rule="merit"); el = time.perf_counter() - t0
    r = s.get_results(); res[mode] = r
    print(f"{mode}: {1e3 * el / 20:.3f} ms per iteration, last step {st[-1]:.2e}, status {np.bincount(r['status'], minlength=3).tolist()}, riccati/polish its max {r['polish_iters'].max()}")
    s.close()
print("u difference between the two:", np.abs(res["condensed"]["u"] - res["structured"]["u"]).max())
s3 = capi.Solver(n, m, N, 4)
try:
    s3.sqp_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, x_ref, u_ref, Q, R, None, P, -np.ones(m), np.ones(m), act="tanh", qp_solver="structured",
                     xmin=-np.ones(4), xmax=np.ones(4))
    print("state rows + structured: NO ERROR (unexpected)")
except capi.AlmpcError as e:
    print("state rows + structured:", e)
