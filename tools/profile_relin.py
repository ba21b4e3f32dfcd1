"""BASELINE configs[3] alone (1024 x Fnn 4-2-16x2 relu, N = 20, re-linearised every step): the command profiled for its kernels.
   python tools/profile_relin.py [steps]      env ALMPC_NO_INST_WAVE=1: the two-launch step"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, almpc_loader, importlib
pkg = almpc_loader.load_package(); capi = pkg._capi
wl = importlib.import_module(pkg.__name__ + ".workloads")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
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
o3 = capi.default_opts(max_iter=int(os.environ.get('K', '25')), check_every=int(os.environ.get('K', '25')), polish=int(os.environ.get('POLISH', '1')))
for _ in range(5): s3.relin_fnn_step(o3)
t0 = time.perf_counter()
for _ in range(steps): s3.relin_fnn_step(o3, sync=False)
s3.synchronize()
el = time.perf_counter() - t0
r = s3.get_results(want=("status", "polish_iters", "iters"))
print("ms/step %.4f  status %s  polish iters mean %.2f max %d  admm iters mean %.1f" % (1e3 * el / steps, np.bincount(r["status"], minlength=3).tolist(),
      r["polish_iters"].mean(), r["polish_iters"].max(), r["iters"].mean()))
s3.close()
