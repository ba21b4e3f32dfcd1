import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo, c_oracle as co
capi = almpc_loader.load_package()._capi
p = mo.quadrotor(); des = mo.design_shared(p)
X0 = mo.quadrotor_x0_batch(48, 10.0, first_instance=300)
s = capi.Solver(12, 4, 30, 48); s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max); s.set_reference(p.x_ref, p.u_ref)
s.update_initialization(X0); s.calculate(); r = s.get_results()
c = co.step_batch(p, des, X0)
u = c['u']; na = (np.isclose(u, p.u_min[None,:,None])|np.isclose(u, p.u_max[None,:,None])).sum((1,2))
k0=[]
for i in range(48):
    a = mo.admm_box(des["Hs"], des["Fs"] @ X0[i], des["lo"], des["hi"], Minv=des["Minv"], unscale=des["d"], max_iter=50)
    k0.append(int((((a['y']<0)&(a['z']<=des['lo']))|((a['y']>0)&(a['z']>=des['hi']))).sum()))
for i in range(48):
    print(i, "status", r['status'][i], "pit gpu", r['polish_iters'][i], "pit oracle", c['polish_iters'][i], "k0", k0[i], "final active", na[i], "err", np.abs(r['u'][i]-c['u'][i]).max())
