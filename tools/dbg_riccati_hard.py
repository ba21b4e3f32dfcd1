import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, almpc_loader, mpc_oracle as mo, bench
capi = almpc_loader.load_package()._capi
p = mo.quadrotor(N=50); B=4096
X0 = bench.make_x0(mo, 0, B)
s = capi.Solver(12,4,50,B,structured=True)
s.design_shared(p.A,p.B,p.Q,p.R,p.S,None,p.u_min,p.u_max); s.set_reference(p.x_ref,p.u_ref); s.update_initialization(X0); s.calculate()
r = s.get_results(want=("polish_iters","u")); s.close()
pi = r["polish_iters"]; o = np.argsort(-pi)[:10]
print("top", o, pi[o]); print("hist", np.percentile(pi,[50,90,99,99.9]))
for i in o[:3]:
    u = r["u"][i]; act = ((u<=p.u_min[:,None])|(u>=p.u_max[:,None]))
    print(i, "active per input", act.sum(axis=1), "stages active any", np.nonzero(act.any(axis=0))[0][:40])
np.save("/root/repo/gpurun_out/ric_hard_idx.npy", o)
