"""Pins of the LMPC and MHE oracles - and of the HIP paths behind them - to numbers the REFERENCE holds.

The reference's own LMPC / MHE tests assert no number, but two known-answer tests of neighbouring components pin the
same mathematics:

* tests/test_LQR.py:328,342 - the LQR gain of the 3-state / 2-input discrete model of :245-251 after `horizon = 5` Riccati steps
  from P = Q (controller/lqr.py:236-245):  K = [[1.39207671, 1.35221712, 0], [0, 0, .61802575]]  (p = 1) and
  [[0, 0, 0], [0, 0, .61802575]]  (p = 0).  The first move of an UNCONSTRAINED linear MPC with stage weights Q, R, terminal weight
  P = Q and horizon N = 6 is  u_0 = -K x_0  with exactly that gain (five backward steps from P_6 = Q give P_1; K_0 is formed from
  P_1) - for the LMPC's QP in the reference's layout (mpc.py:2188-2265) solved by the oracle and by the HIP QP kernels.
* tests/test_KFs.py:298-316, :488-522 - the Kalman filter on the 2-state linear chain of :247-255 (predict / update / one step;
  oracle/kf.py and the HIP filter reproduce them: tests/test_oracle_kf.py, tests/test_kf_gpu.py).  A moving-horizon estimator
  WITHOUT state noise on a linear model is that filter with Q = 0: arrival weight P0^-1, stage weight R^-1, and - quirk Q7 - no
  stage term at k = 0 and the one-step-ahead state x_N returned (mhe.py:738-748, :381-384), i.e. updates with y_1 .. y_{N-1} and a
  final prediction.  WITH state noise the reference leaves w_0 unpenalised (same lines): the arrival term decouples and the
  estimate is the filter started without prior information at x_1 - compared with the filter started from P = 1e9 I (1e-5).

The filter the estimators are compared with is the one the reference's numbers pin; the chain reference KAT -> filter -> estimator
has no free parameter."""
import numpy as np
import pytest

from oracle import kf as okf
from oracle import models as omodels
from oracle.lmpc import LmpcProblem, lmpc_optimize
from oracle.mhe_gen import MheGenIpm, MheGenProblem
from oracle.nmpc import IpmOptions

K_P1 = np.array([[1.39207671, 1.35221712, 0.], [0., 0., .61802575]])        # tests/test_LQR.py:342
K_P0 = np.array([[0., 0., 0.], [0., 0., .61802575]])                         # tests/test_LQR.py:328


def lqr_model(p, dt=1.):
    """tests/test_LQR.py:245-251: x+ = x + dt (2 y + p u), y+ = y - dt x, z+ = z + dt w"""
    A = np.array([[1., 2. * dt, 0.], [-dt, 1., 0.], [0., 0., 1.]])
    B = np.array([[p * dt, 0.], [0., 0.], [0., dt]])
    return A, B


def riccati_gain(A, B, Q, R, steps):
    """controller/lqr.py:236-245 with N = 0: `steps` backward steps from P = Q, then the gain"""
    P = Q.copy()
    for _ in range(steps):
        APB = A.T @ P @ B
        P = A.T @ P @ A - np.linalg.solve((R + B.T @ P @ B).T, APB.T).T @ (B.T @ P @ A) + Q
    return np.linalg.solve(R + B.T @ P @ B, B.T @ P @ A)


X0 = np.array([[1., 0., 1.], [-.7, .4, 2.], [.3, -1.2, -.5], [2., 1., -1.]])


@pytest.mark.parametrize('p,K', [(1., K_P1), (0., K_P0)])
def test_the_reference_lqr_gain_is_five_riccati_steps_from_q(p, K):
    A, B = lqr_model(p)
    np.testing.assert_allclose(riccati_gain(A, B, np.eye(3), np.eye(2), 5), K, rtol=1e-7, atol=1e-9)    # the reference's own tolerance


@pytest.mark.parametrize('p,K', [(1., K_P1), (0., K_P0)])
def test_oracle_lmpc_first_move_is_the_reference_lqr_gain(p, K):
    """oracle/lmpc.py (QP of mpc.py:2188-2265, corrected input block: nx > 1) with N = 6, Q = I, R = I, P = Q, no active bounds"""
    A, B = lqr_model(p)
    pb = LmpcProblem(A=A, B=B, N=6, Q=np.eye(3), R=np.eye(2), P=np.eye(3), kron_bug=False)
    ref = lmpc_optimize(pb, X0)
    assert np.all(ref['status'] == 1)
    np.testing.assert_allclose(ref['u'], -(X0 @ K.T), rtol=1e-7, atol=2e-8)


# ---- the Kalman filter of tests/test_KFs.py on the 2-state chain, and the estimators that must agree with it ----
KF_DT, KF_P = 1., [.5, .4]
KF_R, KF_X0 = .064, np.array([.8, 0.])


def kf_data(N, seed=4):
    """inputs and measurements of a window: the chain driven by u, y = x_2 + noise"""
    rng = np.random.default_rng(seed)
    model = omodels.get('linear2').discretize(1)
    u = .8 + .2 * rng.uniform(-1, 1, (N, 1))
    x = np.array([[1., .2]])
    xs = [x[0]]
    for k in range(N):
        x = model.f(x, u[k:k + 1], np.array([KF_P]), KF_DT)
        xs.append(x[0])
    y = np.array([s[1] for s in xs])[:, None] + .05 * rng.standard_normal((N + 1, 1))
    return u, y


def kalman_window(u, y, N, Q, P0, start=0):
    """the filter the reference's KATs pin (oracle/kf.py) over a window: from (x_start, P0) predict / update with y_k for
    k = start + 1 .. N - 1, then one prediction: the state the estimator returns (x_N)"""
    model = omodels.get('linear2').discretize(1)
    p = np.array([KF_P])
    xP = okf.pack(KF_X0[None, :], np.asarray(P0, dtype=float)[None, :, :])
    for k in range(start + 1, N):
        xP, _ = okf.kf_step(model, xP, y[k][None, :], u[k - 1][None, :], p, Q, KF_R, KF_DT)
    xP = okf.kf_predict(model, xP, u[N - 1][None, :], p, Q, KF_DT)
    return okf.unpack(xP)[0][0]


def test_the_filter_used_here_reproduces_the_reference_kat():
    """tests/test_KFs.py:298-316 (one step: [1.19614861, .39044856]) - the anchor of the chain"""
    model = omodels.get('linear2').discretize(1)
    xP, _ = okf.kf_step(model, okf.pack(KF_X0[None, :], np.eye(2)[None]), np.array([[.3894626]]), np.array([[.8]]), np.array([KF_P]),
                        [.01, .01], KF_R, KF_DT)
    np.testing.assert_allclose(okf.unpack(xP)[0][0], [1.19614861, .39044856], rtol=1e-7)


@pytest.mark.parametrize('N', [2, 5])
def test_oracle_mhe_without_state_noise_is_the_kalman_filter_with_q_zero(N):
    u, y = kf_data(N)
    pb = MheGenProblem(omodels.get('linear2'), KF_DT, N, degree=0, order=1, noise=False, Wx=[1., 1.], Wy=[1. / KF_R])
    ipm = MheGenIpm(pb, IpmOptions(tol=1e-12))
    res = ipm.solve(KF_X0[None, :], np.zeros((1, 0)), np.array([KF_P]), u[None, :, :], y[None, :N, :])
    assert res['status'][0] == 1
    np.testing.assert_allclose(res['x_opt'][0], kalman_window(u, y, N, 0., np.eye(2)), rtol=1e-8, atol=1e-10)


def test_oracle_mhe_with_state_noise_is_the_filter_without_prior_at_x1():
    """w_0 carries no cost (mhe.py:742-748): x_1 is free of the arrival term; W_w = Q^-1, W_y = R^-1"""
    from oracle.mhe import MheIpm, MheProblem
    N, Q = 6, .01
    u, y = kf_data(N)
    pb = MheProblem(omodels.get('linear2'), KF_DT, N, order=1, Wx=[1., 1.], Wy=[1. / KF_R], Ww=[1. / Q, 1. / Q])
    res = MheIpm(pb, IpmOptions(tol=1e-12)).solve(KF_X0[None, :], np.array([KF_P]), u[None, :, :], y[None, :N, :])
    assert res['status'][0] == 1
    # the filter from "no information" at x_1: start state irrelevant, covariance 1e9 I; its first step is the update with y_1
    model = omodels.get('linear2').discretize(1)
    p = np.array([KF_P])
    xP = okf.pack(np.zeros((1, 2)), 1e9 * np.eye(2)[None])
    xP = okf.kf_update(model, xP, y[1][None, :], u[0][None, :], p, KF_R, KF_DT)[0]
    for k in range(2, N):
        xP, _ = okf.kf_step(model, xP, y[k][None, :], u[k - 1][None, :], p, [Q, Q], KF_R, KF_DT)
    xN = okf.unpack(okf.kf_predict(model, xP, u[N - 1][None, :], p, [Q, Q], KF_DT))[0][0]
    np.testing.assert_allclose(res['x_opt'][0], xN, rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------------------------------------------------------------------
# the HIP paths against the same numbers
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('p,K', [(1., K_P1), (0., K_P0)])
def test_hip_lmpc_first_move_is_the_reference_lqr_gain(p, K):
    """`LMPC.optimize` (hilo_qp_solve: the stage-structured kernel for this block layout) - unconstrained, so u_0 = -K x_0"""
    from hilo_mpc_amd import LMPC, Model
    A, B = lqr_model(p)
    mpc = LMPC(Model('lti', A=A, B=B).setup(dt=1.))
    mpc.Q, mpc.R, mpc.P = np.eye(3), np.eye(2), np.eye(3)
    mpc.horizon = 6
    mpc.setup(kron_variant='corrected')
    u = mpc.optimize(X0)
    assert np.all(mpc.solver_status_code == 1)
    np.testing.assert_allclose(u, -(X0 @ K.T), rtol=1e-7, atol=2e-8)


def _chain_model():
    from hilo_mpc_amd import Model
    m = Model()
    x = m.set_dynamical_states(['x_1', 'x_2'])
    u = m.set_inputs(['u'])
    k = m.set_parameters(['k_1', 'k_2'])
    m.set_dynamical_equations([-k[0] * x[0] + u[0], k[0] * x[0] - k[1] * x[1]])
    m.set_measurement_equations([x[1]])
    return m


@pytest.mark.gpu
@pytest.mark.parametrize('N', [2, 5])
def test_hip_mhe_without_state_noise_is_the_kalman_filter_with_q_zero(N):
    from hilo_mpc_amd import MHE
    u, y = kf_data(N)
    m = _chain_model().discretize('erk', order=1).setup(dt=KF_DT)
    mhe = MHE(m)
    mhe.quad_arrival_cost.add_states(weights=[1., 1.], guess=list(KF_X0))
    mhe.quad_stage_cost.add_measurements(weights=[1. / KF_R])
    mhe.horizon = N
    mhe.set_box_constraints(p_lb=KF_P, p_ub=KF_P)
    mhe.set_initial_guess(x_guess=list(KF_X0))
    mhe.setup(options={'integration_method': 'discrete'}, nlp_opts={'ipopt.tol': 1e-12})
    x_est = None
    for k in range(N):
        mhe.add_measurements(y[k], u_meas=u[k])
        x_est, _ = mhe.estimate()
    assert np.all(mhe.solver_status_code == 1)
    np.testing.assert_allclose(x_est.cpu().numpy().reshape(-1), kalman_window(u, y, N, 0., np.eye(2)), rtol=1e-8, atol=1e-10)


@pytest.mark.gpu
def test_hip_mhe_with_state_noise_is_the_filter_without_prior_at_x1():
    """the estimator of configuration 3's kind (state noise, pre-discretised model, `discrete`): w_0 carries no cost, so the window's
    estimate is the reference-pinned filter started without information at x_1 (P = 1e9 I: 1e-5)"""
    from hilo_mpc_amd import MHE
    N, Q = 6, .01
    u, y = kf_data(N)
    m = _chain_model().discretize('erk', order=1).setup(dt=KF_DT)
    mhe = MHE(m)
    mhe.quad_arrival_cost.add_states(weights=[1., 1.], guess=list(KF_X0))
    mhe.quad_stage_cost.add_measurements(weights=[1. / KF_R])
    mhe.quad_stage_cost.add_state_noise(weights=[1. / Q, 1. / Q])
    mhe.horizon = N
    mhe.set_box_constraints(p_lb=KF_P, p_ub=KF_P)
    mhe.set_initial_guess(x_guess=list(KF_X0))
    mhe.setup(options={'integration_method': 'discrete'}, nlp_opts={'ipopt.tol': 1e-12})
    x_est = None
    for k in range(N):
        mhe.add_measurements(y[k], u_meas=u[k])
        x_est, _ = mhe.estimate()
    assert np.all(mhe.solver_status_code == 1)
    model = omodels.get('linear2').discretize(1)
    p = np.array([KF_P])
    xP = okf.pack(np.zeros((1, 2)), 1e9 * np.eye(2)[None])
    xP = okf.kf_update(model, xP, y[1][None, :], u[0][None, :], p, KF_R, KF_DT)[0]
    for k in range(2, N):
        xP, _ = okf.kf_step(model, xP, y[k][None, :], u[k - 1][None, :], p, [Q, Q], KF_R, KF_DT)
    xN = okf.unpack(okf.kf_predict(model, xP, u[N - 1][None, :], p, [Q, Q], KF_DT))[0][0]
    np.testing.assert_allclose(x_est.cpu().numpy().reshape(-1), xN, rtol=1e-5, atol=1e-6)
