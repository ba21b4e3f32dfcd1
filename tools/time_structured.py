"""Time the stage-wise dual solve (k_sdual) on structured handles: quadrotor, 4096 instances, mixed amplitudes.
python tools/time_structured.py [N ...]   (default 50 30)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
b = 4096
X0 = np.concatenate([mo.quadrotor_x0_batch(b // 4 if a != 1.0 else b // 2, a, first_instance=k * b) for k, a in enumerate((0.3, 1.0, 3.0))])[:b]
XMAX = np.array([3, 3, 3, 1.5, 1.5, 1.5, 0.3, 0.3, 0.3, 1.0, 1.0, 1.0])
Ns = [int(a) for a in sys.argv[1:]] or [50, 30]
for N in Ns:
    q = mo.quadrotor(N)
    for name, kw in (("box-only", {}), ("state box", dict(x_min=-XMAX, x_max=XMAX)), ("equality", dict(terminal="equality")),
                     ("box+eq", dict(x_min=-XMAX, x_max=XMAX, terminal="equality")), ("S=5", dict(s=5.0))):
        p = mo.make_problem(q.A, q.B, N, q.u_min, q.u_max, **kw)
        X = np.clip(X0, -0.99 * XMAX, 0.99 * XMAX) if p.x_min is not None else X0
        s = capi.Solver(12, 4, N, b, structured=True)
        s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=p.x_min, xmax=p.x_max, terminal=p.terminal)
        s.set_reference(p.x_ref, p.u_ref); s.update_initialization(X)
        s.calculate()
        best = float("inf")
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(3): s.calculate(sync=False)
            s.synchronize()
            best = min(best, (time.perf_counter() - t0) / 3)
        r = s.get_results(want=("status", "polish_iters"))
        print(f"N {N} {name:10s}: {1e3*best:8.3f} ms/step, status {np.bincount(r['status'], minlength=4).tolist()}, changes mean {r['polish_iters'].mean():.1f} max {r['polish_iters'].max()}", flush=True)
        s.close()
