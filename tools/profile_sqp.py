"""SQP iterations of BASELINE configs[4] alone, for rocprofv3 --kernel-trace --stats (see tools/README.md)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import almpc_loader
pkg = almpc_loader.load_package()
capi = pkg._capi
import mpc_oracle as mo

bq = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Nq = int(sys.argv[2]) if len(sys.argv) > 2 else 50
its = int(sys.argv[3]) if len(sys.argv) > 3 else 20
max_iter = int(sys.argv[4]) if len(sys.argv) > 4 else 25
f = mo.synthetic_fnn(act="tanh")
nq, mq = 4, 2
xr = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, Nq + 1)); ur = np.tile(np.array([0.1, -0.2])[:, None], (1, Nq))
X0 = xr[:, 0][None, :] + 0.6 * mo.splitmix_normal(0x5EED0005, 0, bq, nq)
Al, Bl = f.jacobian(xr[:, -1], ur[:, -1])
P = mo.dare(Al, Bl, 100.0 * np.eye(nq), 0.1 * np.eye(mq))
s = capi.Solver(nq, mq, Nq, bq)
# env QP=structured: every iteration's QP in its stage-wise form (k_sgains + k_sdual) instead of the condensed design
s.sqp_fnn_setup(f.W_in, f.W_h, f.b_h, f.W_out, xr, ur, 100.0 * np.eye(nq), 0.1 * np.eye(mq), None, P, -np.ones(mq), np.ones(mq), act="tanh",
                qp_solver=os.environ.get("QP", "condensed"))
import time
for rep in range(3):
    s.sqp_fnn_start(X0)
    t0 = time.perf_counter()
    st, de = s.sqp_fnn_iterate(its, opts=capi.default_opts(max_iter=max_iter, check_every=max_iter))
    r = s.get_results(want=('polish_iters', 'iters', 'status'))
    print('polish iters mean/max', r['polish_iters'].mean(), r['polish_iters'].max(), 'admm iters', r['iters'].mean(), 'status', np.bincount(r['status']))
    print(f"rep {rep}: {1e3 * (time.perf_counter() - t0) / its:.3f} ms / iteration; step {st[-1]:.2e} defect {de[-1]:.2e}")
print("steps", np.array2string(st, precision=2))
s.close()
