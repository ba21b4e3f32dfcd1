"""configs[3] pipeline: which instance sets the step time -- polish iterations and active bounds of the slowest ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, almpc_loader, importlib
pkg = almpc_loader.load_package(); capi = pkg._capi
wl = importlib.import_module(pkg.__name__ + ".workloads")
b3, n3, m3, N3 = 1024, 4, 2, 20
W_in, W_h, b_h, W_out = wl.synthetic_fnn_weights(n3, m3)
A0, _ = capi.fnn_linearize(W_in, W_h, b_h, W_out, np.zeros((1, n3)), np.zeros((1, m3)), act="relu")
W_out = wl.scale_to_radius(W_out, A0[0])
xr3 = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N3 + 1)); ur3 = np.tile(np.array([0.1, -0.2])[:, None], (1, N3))
X03 = xr3[:, 0][None, :] + wl.splitmix_normal(0x5EED0004, 0, b3, n3)
Q3, R3 = 100.0 * np.eye(n3), 0.1 * np.eye(m3)
Al3, Bl3 = capi.fnn_linearize(W_in, W_h, b_h, W_out, xr3[:, -1][None, :], ur3[:, -1][None, :], act="relu")
P3 = capi.dare(Al3[0], Bl3[0], Q3, R3)
s3 = capi.Solver(n3, m3, N3, b3)
s3.relin_fnn_setup(W_in, W_h, b_h, W_out, xr3, ur3, Q3, R3, None, P3, -np.ones(m3), np.ones(m3), act="relu")
s3.update_initialization(X03)
s3.relin_fnn_step(capi.default_opts())
r = s3.get_results(want=("u", "status", "polish_iters", "iters"))
act = (np.abs(np.abs(r["u"]) - 1.0) < 1e-12).reshape(b3, -1).sum(axis=1)
order = np.argsort(-r["polish_iters"])[:8]
print("polish iterations / active bounds of the slowest instances:", [(int(i), int(r["polish_iters"][i]), int(act[i])) for i in order])
print("active bounds: mean %.1f, max %d, instances with more than 32: %d" % (act.mean(), act.max(), (act > 32).sum()))
s3.close()
