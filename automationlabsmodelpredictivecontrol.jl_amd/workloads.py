"""Synthetic workloads of BASELINE.json: the inputs bench.py and __graft_entry__.smoke() feed to the
library.  Product-side code: nothing here touches oracle/ (the oracle has its own copies of the same definitions;
tests/test_host_logic.py checks that the two produce identical arrays)."""
import math
import types

import numpy as np

_MASK = (1 << 64) - 1
QUADROTOR_X0_SCALE = np.array([1, 1, 1, .5, .5, .5, .1, .1, .1, .1, .1, .1])


def _mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def splitmix_normal(seed: int, first_instance: int, count: int, dim: int):
    """Standard normals (count, dim): one SplitMix64 stream per instance (state0 = mix64(seed + 0x632BE59BD9B4E019 (instance+1))),
    uniforms = top 53 bits, Box-Muller pairs (cos branch first).  Depends only on (seed, instance index): a shard generates its
    own slice of the batch."""
    with np.errstate(over="ignore"):
        inst = np.arange(first_instance, first_instance + count, dtype=np.uint64)
        state = _mix64(np.uint64(seed & _MASK) + np.uint64(0x632BE59BD9B4E019) * (inst + np.uint64(1)))
        npair = (dim + 1) // 2
        out = np.empty((count, 2 * npair))
        gamma = np.uint64(0x9E3779B97F4A7C15)
        for j in range(npair):
            state = state + gamma
            u1 = (_mix64(state) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
            state = state + gamma
            u2 = (_mix64(state) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
            r = np.sqrt(-2.0 * np.log(1.0 - u1))
            out[:, 2 * j] = r * np.cos(2.0 * np.pi * u2)
            out[:, 2 * j + 1] = r * np.sin(2.0 * np.pi * u2)
    return out[:, :dim]


def quadrotor_model(Ts=0.1, mass=0.5, J=(4e-3, 4e-3, 8e-3), g=9.81):
    """Hover-linearised quadrotor, states [p(3), v(3), (phi,theta,psi), omega(3)], inputs [dT, tau_x, tau_y, tau_z]; exact
    zero-order hold via expm([[Ac,Bc],[0,0]] Ts) (configs[1])."""
    import scipy.linalg as sla
    Ac = np.zeros((12, 12))
    Bc = np.zeros((12, 4))
    Ac[0:3, 3:6] = np.eye(3)
    Ac[3, 7] = g
    Ac[4, 6] = -g
    Ac[6:9, 9:12] = np.eye(3)
    Bc[5, 0] = 1.0 / mass
    Bc[9, 1] = 1.0 / J[0]
    Bc[10, 2] = 1.0 / J[1]
    Bc[11, 3] = 1.0 / J[2]
    M = np.zeros((16, 16))
    M[:12, :12] = Ac
    M[:12, 12:] = Bc
    E = sla.expm(M * Ts)
    return E[:12, :12].copy(), E[:12, 12:].copy()


def quadrotor(N=30):
    """configs[1]/[2] controller data: Q = 100 I, R = 0.1 I, S = 0, input box [-2,3] x [+-0.05] x [+-0.05] x [+-0.02], references 0
    (deviation coordinates); the terminal weight is left to the library (P = DARE)."""
    A, B = quadrotor_model()
    return types.SimpleNamespace(A=A, B=B, N=N, n=12, m=4, Q=100.0 * np.eye(12), R=0.1 * np.eye(4), S=np.zeros((4, 4)),
                                 u_min=np.array([-2.0, -0.05, -0.05, -0.02]), u_max=np.array([3.0, 0.05, 0.05, 0.02]),
                                 x_ref=np.zeros((12, N + 1)), u_ref=np.zeros((4, N)))


def quadrotor_x0_batch(batch, amplitude=1.0, seed=0x5EED0002, first_instance=0):
    """x0_i = s diag(1,1,1,.5,.5,.5,.1 x6) xi_i, xi_i ~ N(0, I12); (batch, 12), instance-major."""
    xi = splitmix_normal(seed, first_instance, batch, 12)
    return np.ascontiguousarray(amplitude * xi * QUADROTOR_X0_SCALE[None, :])


def synthetic_fnn_weights(n=4, m=2, H=16, L=2, seed=0x5EED0004):
    """configs[3]/[4] network in the layout the reference reads from Flux.params: weights ~ U(-1,1)/sqrt(fan_in) from the same
    generator (one stream per tensor).  Returns (W_in, W_h, b_h, W_out) BEFORE the output scaling; scale_to_radius does that."""
    def uni(stream, shape, fan_in):
        with np.errstate(over="ignore"):
            cnt = int(np.prod(shape))
            state = _mix64(np.uint64(seed & _MASK) + np.uint64(0x632BE59BD9B4E019) * np.uint64(stream + 1))
            out = np.empty(cnt)
            gamma = np.uint64(0x9E3779B97F4A7C15)
            for i in range(cnt):
                state = state + gamma
                out[i] = float(_mix64(state) >> np.uint64(11)) * 2.0 ** -53
        return (2.0 * out.reshape(shape) - 1.0) / math.sqrt(fan_in)
    W_in = uni(0, (H, n + m), n + m)
    W_h = [uni(1 + 2 * j, (H, H), H) for j in range(L)]
    b_h = [0.1 * uni(2 + 2 * j, (H,), H) * math.sqrt(H) for j in range(L)]
    W_out = uni(100, (n, H), H)
    return W_in, W_h, b_h, W_out


def scale_to_radius(W_out, A0, radius=0.95):
    """W_out rescaled so that the Jacobian at the origin (A0, computed with the unscaled W_out) has the given spectral radius."""
    return W_out * (radius / max(1e-12, float(np.max(np.abs(np.linalg.eigvals(A0))))))


def qtp_fixture(N=5):
    """The reference's own test size (test/computation_mpc_test.jl:981-1054): quadruple-tank model n 4, m 2 as its test fixture
    test/models_saved/linear_regressor_train_result.jls holds it (Float32 coefficients, widened), N = 5, x_ref 0.65, u_ref 1.2,
    u in [0, 4] x [0, 3.26], Q = 100 I, R = 0.1 I, no state box.  The matrices are DATA of that fixture (the oracle decodes the
    same values from the blob: tests/test_host_logic.py keeps the two in step)."""
    A = np.array([[0.9680719971656799, -0.004229385405778885, 0.020768633112311363, 0.0012167866807430983],
                  [-0.008298816159367561, 0.9640175700187683, 0.0032141467090696096, 0.018493792042136192],
                  [-0.01220669038593769, -0.005800441838800907, 0.9852687120437622, 0.0007298348937183619],
                  [-0.0077879056334495544, -0.011747360229492188, 0.004276304971426725, 0.9848219752311707]])
    B = np.array([[0.00622002687305212, 4.157558942097239e-05], [9.120091999648139e-05, 0.00806921161711216],
                  [-0.00022227228328119963, 0.012224327772855759], [0.01440966036170721, -0.00029592248029075563]])
    return types.SimpleNamespace(A=A, B=B, N=N, n=4, m=2, Q=100.0 * np.eye(4), R=0.1 * np.eye(2), S=np.zeros((2, 2)),
                                 u_min=np.array([0.0, 0.0]), u_max=np.array([4.0, 3.26]),
                                 x_ref=0.65 * np.ones((4, N + 1)), u_ref=1.2 * np.ones((2, N)))
