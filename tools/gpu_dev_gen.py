"""Developer check of the general-constraint path (state box / terminal equality) vs the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, almpc_loader, mpc_oracle as mo
capi = almpc_loader.load_package()._capi
def run(p, X0, label, **optkw):
    X0 = np.atleast_2d(X0); b = len(X0)
    s = capi.Solver(p.n, p.m, p.N, b)
    s.design_shared(p.A, p.B, p.Q, p.R, p.S, None, p.u_min, p.u_max, xmin=p.x_min, xmax=p.x_max, terminal=p.terminal)
    s.set_reference(p.x_ref, p.u_ref); s.update_initialization(X0); s.calculate(capi.default_opts(**optkw)); r = s.get_results(); s.close()
    errs = []; feas = []
    for i in range(b):
        try:
            e = mo.solve_mpc_exact(p, X0[i], return_info=True); feas.append(True)
            errs.append((np.abs(r['u'][i]-e['u']).max(), np.abs(r['x'][i]-e['x']).max(), e['info']['n_active_state']))
        except ValueError:
            feas.append(False); errs.append((np.nan, np.nan, -1))
    errs = np.array(errs); feas = np.array(feas)
    print(f"[{label}] status {np.bincount(r['status'], minlength=4)} oracle-feasible {feas.sum()}/{b}; status==3 matches infeasible: {np.array_equal(r['status']==3, ~feas)}")
    if feas.any():
        ok = feas & (r['status'] == 0)
        print(f"   solved&feasible {ok.sum()}: max|u-u*| {np.nanmax(errs[ok,0]):.2e} max|x-x*| {np.nanmax(errs[ok,1]):.2e}  active state rows max {int(np.nanmax(errs[ok,2]))}  polish its mean {r['polish_iters'][ok].mean():.1f} max {r['polish_iters'][ok].max()}")
        bad = feas & (r['status'] != 0)
        if bad.any(): print("   NOT solved though feasible:", np.flatnonzero(bad)[:10], r['status'][bad][:10])
p = mo.make_problem([[1.,1.],[0.,1.]], [[0.5],[1.]], 10, [-1.],[1.], x_min=[-10., -0.8], x_max=[10., 0.8])
run(p, np.array([[5.,0.],[-6.,0.5],[1.,0.],[0.,0.],[20.,0.],[3.,0.7]]), "DI state box")
p2 = mo.make_problem([[1.,1.],[0.,1.]], [[0.5],[1.]], 10, [-1.],[1.], terminal="equality")
run(p2, np.array([[2.,0.],[1.,-0.5],[30.,0.],[0.,0.]]), "DI terminal equality")
p3 = mo.make_problem([[1.,1.],[0.,1.]], [[0.5],[1.]], 10, [-1.],[1.], x_min=[-10., -0.8], x_max=[10., 0.8], terminal="equality")
run(p3, np.array([[2.,0.],[1.,-0.5],[3.,0.5],[0.,0.]]), "DI box + equality")
q = mo.quadrotor(); xm = np.array([50,50,50, 1.0,1.0,1.0, 0.3,0.3,0.3, 2,2,2.])
q2 = mo.make_problem(q.A,q.B,30,q.u_min,q.u_max,x_min=-xm,x_max=xm)
run(q2, mo.quadrotor_x0_batch(64, 1.0), "quadrotor state box s=1")
run(q2, mo.quadrotor_x0_batch(64, 0.5), "quadrotor state box s=0.5")
q3 = mo.make_problem(q.A,q.B,30,q.u_min,q.u_max,terminal="equality")
run(q3, mo.quadrotor_x0_batch(32, 0.1), "quadrotor terminal equality s=0.1")
