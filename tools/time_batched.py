"""Time the per-instance-model path (almpc_design_batched + step): python tools/time_batched.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi


def run(name, As, Bs, N, umin, umax, X0, P, K=8, reps=20):
    b, n, m = As.shape[0], As.shape[1], Bs.shape[2]
    s = capi.Solver(n, m, N, b, timing=True)
    kw = dict(rho=30.0, rho_profile="stiffness")
    t0 = time.perf_counter(); s.design_batched(As, Bs, 100 * np.eye(n), 0.1 * np.eye(m), None, P, umin, umax, **kw); t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(5): s.design_batched(As, Bs, 100 * np.eye(n), 0.1 * np.eye(m), None, P, umin, umax, **kw)
    t_design = (time.perf_counter() - t0) / 5
    s.update_initialization(X0)
    opts = capi.default_opts(rho=30.0, max_iter=K, check_every=K)
    for _ in range(3): s.calculate(opts)
    s.timing_reset(reps)
    t0 = time.perf_counter()
    for _ in range(reps): s.calculate(opts, sync=False)
    s.synchronize()
    t_step = (time.perf_counter() - t0) / reps
    ts = s.timing_summary()
    r = s.get_results(want=("status", "polish_iters"))
    nz = m * N
    byt = b * 8 * (2 * nz * (16 * ((nz + 15) // 16)))  # Minv_i + (rows of) G_i upper bound
    print(f"{name}: batch {b} n {n} m {m} N {N}: design first {1e3*t_first:.2f} ms, re-design {1e3*t_design:.2f} ms, step {1e6*t_step:.1f} us "
          f"(admm {1e3*ts['admm_ms']/ts['steps']:.1f} us, polish {1e3*ts['polish_ms']/ts['steps']:.1f} us) -> {b/t_step/1e6:.2f} M instance-steps/s, "
          f"Minv+G bytes/step {byt/1e6:.0f} MB = {byt/t_step/1e9:.0f} GB/s; status {np.bincount(r['status']).tolist()} polish its mean {r['polish_iters'].mean():.2f} max {r['polish_iters'].max()} top5 {np.sort(r['polish_iters'])[-5:].tolist()}")
    s.close()


rng = np.random.default_rng(0)
# config 4 shape: Fnn linearised per instance
f = mo.synthetic_fnn()
b = 1024
x_ref, u_ref = np.array([0.2, -0.1, 0.05, 0.0]), np.array([0.1, -0.2])
X0 = x_ref[None, :] + mo.splitmix_normal(0x5EED0004, 0, b, 4) * 2.0
t0 = time.perf_counter(); A, B = capi.fnn_linearize(f.W_in, f.W_h, f.b_h, f.W_out, X0, np.repeat(u_ref[None], b, 0), act=f.act); tl = time.perf_counter() - t0
print(f"fnn_linearize (host in/out) {1e3*tl:.2f} ms")
run("config4 (P = DARE per instance on the host)", A, B, 20, [-1, -1], [1, 1], X0, None)
run("config4 (shared P)", A, B, 20, [-1, -1], [1, 1], X0, 100.0 * np.eye(4))
# quadrotor family
b = 4096
p = mo.quadrotor()
As = np.repeat(p.A[None], b, 0) * (1.0 + 0.01 * rng.standard_normal((b, 1, 1)) * 0)  # identical plants: timing only
Bs = np.repeat(p.B[None], b, 0) * (1.0 + 0.05 * rng.standard_normal((b, 1, 1)))
X0 = mo.quadrotor_x0_batch(b, 1.0)
run("quadrotor family (shared P)", As, Bs, 30, p.u_min, p.u_max, X0, p.P)
