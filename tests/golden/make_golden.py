"""Generates tests/golden/*.json with oracle/mpc_oracle.py (exact, KKT-certified optimum).

NOT reference output: the reference (Julia + OSQP/SCIP/Ipopt) cannot run in this image and its tests hold
no numeric golden vectors.  These vectors freeze the oracle so that a later change to
it, or to the HIP path, is caught; the only reference-held data here is linear_regressor_train_result.jls
(copied from the reference's test/models_saved/, 555 bytes) from which (A, B) of the QTP case are decoded.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import mpc_oracle as mo  # noqa: E402


def case(p, X0, name, extra=None):
    out = []
    for x0 in np.atleast_2d(X0):
        e = mo.solve_mpc_exact(p, x0)
        out.append(dict(x0=x0.tolist(), u=e["u"].tolist(), x=e["x"].tolist(),
                        n_active=int(((e["u"] <= p.u_min[:, None] + 0) | (e["u"] >= p.u_max[:, None])).sum())))
    d = dict(name=name, source="oracle/mpc_oracle.py::solve_mpc_exact (KKT-certified); NOT reference output",
             n=p.n, m=p.m, N=p.N, A=p.A.tolist(), B=p.B.tolist(), q=float(p.Q[0, 0]), r=float(p.R[0, 0]), s=float(p.S[0, 0]),
             P=p.P.tolist(), u_min=p.u_min.tolist(), u_max=p.u_max.tolist(), x_ref=p.x_ref[:, 0].tolist(),
             u_ref=p.u_ref[:, 0].tolist(), cases=out)
    if extra:
        d.update(extra)
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(d, f)
    print(name, len(out), "cases")


def sqp_case(name="fnn_sqp"):
    """NLP branch (Fnn as equality constraints): converged Gauss-Newton SQP of the restatement, certified by the NLP's own
    projected-gradient residual."""
    f = mo.synthetic_fnn(act="tanh")
    n, m, N = 4, 2, 20
    x_ref = np.tile(np.array([0.2, -0.1, 0.05, 0.0])[:, None], (1, N + 1)); u_ref = np.tile(np.array([0.1, -0.2])[:, None], (1, N))
    Al, Bl = f.jacobian(x_ref[:, -1], u_ref[:, -1])
    Q, R, S = 100.0 * np.eye(n), 0.1 * np.eye(m), np.zeros((m, m))
    P = mo.dare(Al, Bl, Q, R)
    umin, umax = -np.ones(m), np.ones(m)
    X0 = x_ref[:, 0][None, :] + 0.6 * mo.splitmix_normal(0x5EED000A, 0, 4, n)
    out = []
    for x0 in X0:
        X, U, hist = mo.sqp_fnn(f, x0, x_ref, u_ref, Q, R, S, P, umin, umax, iters=60)
        res = mo.nlp_kkt_residual(f, x0, U, x_ref, u_ref, Q, R, S, P, umin, umax)
        assert res <= 1e-9 and hist[-1][0] <= 1e-9, (res, hist[-1])
        out.append(dict(x0=x0.tolist(), u=U.tolist(), x=X.tolist(), kkt=res, n_active=int(((U <= -1) | (U >= 1)).sum()),
                        cost=mo.nlp_cost_and_gradient(f, x0, U, x_ref, u_ref, Q, R, S, P)[0]))
    d = dict(name=name, source="oracle/mpc_oracle.py::sqp_fnn (60 iterations, exact QP solves), certified by nlp_kkt_residual; "
                               "NOT reference output (the reference gives this NLP to Ipopt)",
             n=n, m=m, N=N, act="tanh", W_in=f.W_in.tolist(), W_h=[w.tolist() for w in f.W_h], b_h=[b.tolist() for b in f.b_h],
             W_out=f.W_out.tolist(), q=100.0, r=0.1, s=0.0, P=P.tolist(), u_min=umin.tolist(), u_max=umax.tolist(),
             x_ref=x_ref[:, 0].tolist(), u_ref=u_ref[:, 0].tolist(), cases=out)
    with open(os.path.join(HERE, name + ".json"), "w") as fh:
        json.dump(d, fh)
    print(name, len(out), "cases, active bounds:", [c["n_active"] for c in out])


if __name__ == "__main__":
    sqp_case()
    case(mo.double_integrator(), np.array([[1.0, 0.0], [5.0, 0.0], [-3.0, 1.0], [0.0, 0.0]]), "double_integrator")
    with open(os.path.join(HERE, "linear_regressor_train_result.jls"), "rb") as f:
        A, B = mo.decode_linear_regressor_fixture(f.read())
    case(mo.qtp_linear_fixture_problem(A, B), np.array([[0.6] * 4, [0.3, 0.9, 0.5, 1.2]]), "qtp_linear",
         extra=dict(scenario="test/computation_mpc_test.jl:981-1054 (first case); second case is an extra x0"))
    q = mo.quadrotor()
    X0 = np.concatenate([mo.quadrotor_x0_batch(6, s, first_instance=100 * k) for k, s in enumerate((0.3, 1.0, 3.0, 10.0))])
    case(q, X0, "quadrotor", extra=dict(amplitudes=[0.3] * 6 + [1.0] * 6 + [3.0] * 6 + [10.0] * 6))
    # S != 0 (input-rate cost) and non-zero references on the double integrator
    p = mo.make_problem([[1.0, 1.0], [0.0, 1.0]], [[0.5], [1.0]], 8, [-1.0], [1.0], x_ref=[0.0, 0.0], u_ref=[0.0], s=2.0)
    case(p, np.array([[2.0, 0.5], [-4.0, 0.0]]), "double_integrator_S")
