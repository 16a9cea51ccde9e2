"""GPU parity: hilo_kf_* (through the host classes and the C ABI) vs the reference's known answers and the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import kf as okf, models as omodels          # noqa: E402
from tests.util import spd_batch                        # noqa: E402

P_BIO = [.15, 303.15, .13, .00025, 15., .14]


def _lin():
    from hilo_mpc_amd import Model
    return Model('linear2').discretize('erk', order=1).setup(dt=1.)


def _toy():
    from hilo_mpc_amd import Model
    return Model('toy1d').setup(dt=1.)


# ---- the reference's own known-answer tests, through the product API ---------------------------------------
def test_kf_predict_update_kat():
    from hilo_mpc_amd import KF
    kf = KF(_lin())
    kf.setup()
    pred = kf.predict(np.array([[.8, 1., 0.], [0., 0., 1.]]), np.array([.8, .5, .4]), .01 * np.eye(2))
    np.testing.assert_allclose(pred, np.array([[1.2, .26, .25], [.4, .25, .62]]))          # test_KFs.py:488-503
    up, yp = kf.update(np.array([[1.2, .26, .25], [.4, .25, .62]]), .322052, np.array([.8, .5, .4]), .064)
    np.testing.assert_allclose(up, np.array([[1.17151023, .16862573, .023391813],
                                             [.32934538, .023391813, .0580117]]), rtol=1e-7)   # :505-522
    np.testing.assert_allclose(yp, np.array([[.4]]))


def test_kf_one_step_kat():
    from hilo_mpc_amd import KF
    kf = KF(_lin())
    kf.setup()
    kf.R = .064
    kf.Q = [.01, .01]
    kf.set_initial_guess([.8, 0.])
    kf.set_initial_parameter_values([.5, .4])
    kf.estimate(y=.3894626, u=.8)
    np.testing.assert_allclose(kf.solution.get_by_id('x:f'), np.array([[1.19614861], [.39044856]]), rtol=1e-7)
    kf2 = KF(_lin())
    kf2.setup()
    kf2.R, kf2.Q = .064, [.01, .01]
    kf2.set_initial_guess([.8, 0.])
    kf2.estimate(y=.3894626, u=.8, p=[.5, .4])                                               # test_KFs.py:318-335
    np.testing.assert_allclose(kf2.solution['x:f'], np.array([[1.19614861], [.39044856]]), rtol=1e-7)


def test_ekf_ukf_one_step_kat():
    from hilo_mpc_amd import EKF, UKF
    for cls, ref in ((EKF, 7.206059), (UKF, 7.2739647)):                                    # test_KFs.py:548-566,606-624
        f = cls(_toy())
        f.setup()
        f.Q = 10.
        f.R = 1.
        f.set_initial_guess(9.)
        f.estimate(y=2.59109)
        np.testing.assert_allclose(f.solution.get_by_id('x:f'), np.array([[ref]]), rtol=1e-7)


def test_ukf_sigma_points_kat():
    from hilo_mpc_amd import UKF, Model
    m = Model('bioreactor3').setup(dt=.1)
    ukf = UKF(m, alpha=1.)
    ukf.setup()
    x = np.array([[299.876], [.217108], [20.]])
    pred = ukf.predict(np.append(x, np.eye(3), axis=1), np.append([.01], P_BIO), np.zeros((3, 3)))
    ref = np.array([[299.925, 0.970453, 1.08254e-06, -2.11206e-05, 299.925, 301.631, 299.925, 299.925, 298.218,
                     299.925, 299.925],
                    [0.219433, 1.08254e-06, 1.02165, -0.0848792, 0.219446, 0.219451, 1.97011, 0.219537, 0.21944,
                     -1.53128, 0.219345],
                    [19.9619, -2.11206e-05, -0.0848792, 1.00427, 19.9618, 19.9617, 19.8164, 21.6914, 19.9618,
                     20.1075, 18.2322]])
    np.testing.assert_allclose(pred, ref, atol=1e-3)                                          # test_KFs.py:716-734
    x = np.array([[299.925], [.219433], [19.9619]])
    P = np.array([[.970453, 1.08254e-06, -2.11206e-05], [1.08254e-06, 1.02165, -.0848792],
                  [-2.11206e-05, -.0848792, 1.00427]])
    X = np.array([[299.925, 301.631, 299.925, 299.925, 298.218, 299.925, 299.925],
                  [.219446, .219451, 1.97011, .219537, .21944, -1.53128, .219345],
                  [19.9618, 19.9617, 19.8164, 21.6914, 19.9618, 20.1075, 18.2322]])
    up, yp = ukf.update(np.concatenate([x, P, X], axis=1), np.array([[300.941], [.245805]]),
                        np.append([.01], P_BIO), np.diag([.25, .01]))
    np.testing.assert_allclose(up, np.array([[300.733, 0.198539, -2.03814e-06, 1.56415e-06],
                                             [0.245549, -2.03814e-06, 0.00990874, -0.000819447],
                                             [19.9597, 1.56415e-06, -0.000819447, 0.997286]]), atol=1e-3)
    np.testing.assert_allclose(yp, np.array([[299.925], [.219434]]), atol=1e-3)             # :736-756


def test_error_behaviour_matches_reference():
    from hilo_mpc_amd import KF, UKF, Model
    kf = KF(_lin())
    with pytest.raises(RuntimeError, match="Kalman filter is not set up. Run KalmanFilter.setup\\(\\) before running "
                                           "simulations."):
        kf.estimate()
    kf.setup()
    with pytest.raises(RuntimeError, match="No initial guess for the states found"):
        kf.estimate()
    kf.set_initial_guess([.8, 0.])
    with pytest.raises(RuntimeError, match="No measurement data supplied."):
        kf.estimate()
    with pytest.raises(ValueError, match="The supplied model is nonlinear"):
        KF(Model('toy1d'))
    with pytest.raises(ValueError, match="alpha needs to lie in the interval"):
        UKF(_toy(), alpha=2.)
    with pytest.raises(ValueError, match="kappa needs to be greater or equal to 0"):
        UKF(_toy(), kappa=-1.)


# ---- batched parity vs the oracle -------------------------------------------------------------------------
def _chemo_batch(B, seed=0):
    rng = np.random.default_rng(seed)
    x = np.array([.1, 40., .5, .2]) * (1 + .1 * rng.uniform(-1, 1, (B, 4)))
    P = spd_batch(rng, B, 4)
    u = rng.uniform(0, .3, (B, 2))
    p = np.tile([100., 4., 1., 0.], (B, 1))
    y = x[:, [0, 2]] + .01 * rng.normal(size=(B, 2))
    return x, P, u, p, y


@pytest.mark.parametrize('B', [1, 129, 1000])
@pytest.mark.parametrize('order', [1, 4])
def test_ekf_step_batch_vs_oracle(B, order):
    from hilo_mpc_amd import EKF, Model
    x, P, u, p, y = _chemo_batch(B)
    om = omodels.get('chemostat4').discretize(order)
    ref, ypr = okf.kf_step(om, okf.pack(x, P), y, u, p, 1e-4, 1e-2, 1.)
    f = EKF(Model('chemostat4').discretize('erk', order=order).setup(dt=1.))
    f.setup()
    f.Q, f.R = 1e-4, 1e-2
    f.set_initial_guess(x, P0=P)
    sol = f.estimate(y=y, u=u, p=p)
    # fp64 tolerance: both sides evaluate the same expression tree up to re-association / FMA contraction
    np.testing.assert_allclose(np.asarray(f.x.cpu()), ref[:, :, 0], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(np.asarray(f.P.cpu()), ref[:, :, 1:], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(sol['y'], ypr, rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize('alpha,rtol', [(1., 1e-10), (1e-3, 2e-5)])
def test_ukf_step_batch_vs_oracle(alpha, rtol):
    """With the default alpha = 1e-3 the sigma-point sums cancel by six digits (centre weight ~ -1e6, kf.py:497),
    so the attainable agreement between two correct fp64 implementations is ~1e-6 relative; alpha = 1 is tight."""
    from hilo_mpc_amd import UKF, Model
    B = 257
    x, P, u, p, y = _chemo_batch(B, seed=1)
    om = omodels.get('chemostat4').discretize(4)
    ref, ypr = okf.ukf_step(om, okf.pack(x, P), y, u, p, 1e-4, 1e-2, 1., alpha=alpha)
    f = UKF(Model('chemostat4').discretize('rk4').setup(dt=1.), alpha=alpha)
    f.setup()
    f.Q, f.R = 1e-4, 1e-2
    f.set_initial_guess(x, P0=P)
    f.estimate(y=y, u=u, p=p)
    scale = np.abs(ref).max()
    np.testing.assert_allclose(np.asarray(f.x.cpu()), ref[:, :, 0], rtol=rtol, atol=rtol * scale)
    np.testing.assert_allclose(np.asarray(f.P.cpu()), ref[:, :, 1:], rtol=rtol, atol=rtol * scale)


def test_ukf_predict_update_tiles_vs_oracle():
    from hilo_mpc_amd import UKF, Model
    B = 70
    x, P, u, p, y = _chemo_batch(B, seed=2)
    om = omodels.get('chemostat4').discretize(4)
    pred_ref = okf.ukf_predict(om, okf.pack(x, P), u, p, 1e-4, 1., alpha=1.)
    f = UKF(Model('chemostat4').discretize('rk4').setup(dt=1.), alpha=1.)
    f.setup()
    up = np.concatenate([u, p], axis=1)
    pred = f.predict(okf.pack(x, P), up, 1e-4)
    assert pred.shape == (B, 4, 14)
    np.testing.assert_allclose(pred, pred_ref, rtol=1e-10, atol=1e-12)
    upd_ref, yp_ref = okf.ukf_update(om, pred_ref, y, u, p, 1e-2, 1., alpha=1.)
    upd, yp = f.update(pred_ref, y, up, 1e-2)
    np.testing.assert_allclose(upd, upd_ref, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(yp, yp_ref, rtol=1e-11)


def test_continuous_ekf_vs_oracle():
    """Continuous model: the reference integrates [x; vec P] with CVODES (kf.py:97-110); the library uses fixed-step
    RK4 (n_sub sub-steps), the oracle scipy DOP853 at rtol 1e-11.  Stated tolerance 1e-7."""
    from hilo_mpc_amd import EKF, Model
    B = 6
    x, P, u, p, y = _chemo_batch(B, seed=3)
    om = omodels.get('chemostat4')
    ref, _ = okf.kf_step(om, okf.pack(x, P), y, u, p, 1e-4, 1e-2, .5)
    f = EKF(Model('chemostat4').setup(dt=.5), n_sub=32)
    f.setup()
    f.Q, f.R = 1e-4, 1e-2
    f.set_initial_guess(x, P0=P)
    f.estimate(y=y, u=u, p=p)
    np.testing.assert_allclose(np.asarray(f.x.cpu()), ref[:, :, 0], rtol=1e-7)
    np.testing.assert_allclose(np.asarray(f.P.cpu()), ref[:, :, 1:], rtol=1e-7, atol=1e-9)


def test_lti_kf_and_shared_inputs():
    from hilo_mpc_amd import KF, Model
    dt = .5
    A = np.array([[1., dt], [0., 1.]])
    Bm = np.array([[dt ** 2 / 2], [dt]])
    Cm = np.array([[1., 0.]])
    f = KF(Model('lti', A=A, B=Bm, C=Cm).setup(dt=dt))
    f.setup()
    f.Q, f.R = 1e-3, 1e-2
    rng = np.random.default_rng(5)
    B = 300
    x = rng.normal(size=(B, 2))
    f.set_initial_guess(x)
    y = rng.normal(size=(B, 1))
    f.estimate(y=y, u=[.3])                      # one shared input for the whole batch (stride 0)
    Pm = A @ np.eye(2) @ A.T + 1e-3 * np.eye(2)
    xm = x @ A.T + .3 * Bm.T
    S = Cm @ Pm @ Cm.T + 1e-2
    K = Pm @ Cm.T / S
    xr = xm + (y - xm @ Cm.T) @ K.T
    np.testing.assert_allclose(np.asarray(f.x.cpu()), xr, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(np.asarray(f.P.cpu())[7], Pm - K @ S @ K.T, rtol=1e-12, atol=1e-14)


def test_kalman_filter_on_a_linear_model_written_as_expressions():
    """kf.py:328-367: the Kalman filter takes any linear model; written as expressions it is compiled at setup and gives the step of the
    matrix form (the double integrator of tests/test_LMPC.py:12-13)."""
    from hilo_mpc_amd import KF, Model
    dt = .5
    m = Model(discrete=True)
    x = m.set_dynamical_states(['x_0', 'x_1'])
    u = m.set_inputs(['u'])
    m.set_dynamical_equations([x[0] + dt * x[1] + dt ** 2 / 2 * u[0], x[1] + dt * u[0]])
    m.set_measurement_equations([x[0]])
    m.setup(dt=dt)
    assert m.is_linear()
    A, Bm, Cm = m.system_matrices()
    np.testing.assert_array_equal(A, [[1., dt], [0., 1.]])
    ref = KF(Model('lti', A=A, B=Bm, C=Cm).setup(dt=dt))
    f = KF(m)
    rng = np.random.default_rng(6)
    x0, y = rng.normal(size=(64, 2)), rng.normal(size=(64, 1))
    for flt in (f, ref):
        flt.setup()
        flt.Q, flt.R = 1e-3, 1e-2
        flt.set_initial_guess(x0)
        flt.estimate(y=y, u=[.3])
        flt.estimate(y=.5 * y, u=[-.1])
    np.testing.assert_allclose(np.asarray(f.x.cpu()), np.asarray(ref.x.cpu()), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(np.asarray(f.P.cpu()), np.asarray(ref.P.cpu()), rtol=1e-12, atol=1e-14)


def test_multi_step_idempotent_layout_and_empty_batch():
    """Size-independent property at a large batch: running predict then update equals the fused step (bit-exact)."""
    from hilo_mpc_amd import EKF, Model
    import torch
    B = 200_003
    x, P, u, p, y = _chemo_batch(B, seed=4)
    f = EKF(Model('chemostat4').discretize('rk4').setup(dt=1.))
    f.setup()
    f.Q, f.R = 1e-4, 1e-2
    up = torch.as_tensor(np.concatenate([u, p], axis=1), device='cuda')
    xP = torch.as_tensor(okf.pack(x, P), device='cuda')
    yd = torch.as_tensor(y, device='cuda')
    pred = f.predict(xP, up)
    upd, yp = f.update(pred, yd, up)
    f.set_initial_guess(torch.as_tensor(x, device='cuda'), P0=torch.as_tensor(P, device='cuda'))
    f.estimate(y=yd, u=torch.as_tensor(u, device='cuda'), p=torch.as_tensor(p, device='cuda'))
    assert torch.equal(f.x, upd[:, :, 0]) and torch.equal(f.P, upd[:, :, 1:])
    # spot-check against the oracle on a slice
    om = omodels.get('chemostat4').discretize(4)
    sl = slice(B - 50, B)
    ref, _ = okf.kf_step(om, okf.pack(x[sl], P[sl]), y[sl], u[sl], p[sl], 1e-4, 1e-2, 1.)
    np.testing.assert_allclose(upd[sl].cpu().numpy(), ref, rtol=1e-10, atol=1e-13)
    e = f.predict(torch.empty(0, 4, 5, dtype=torch.float64, device='cuda'), torch.empty(0, 6, dtype=torch.float64, device='cuda'))
    assert e.shape == (0, 4, 5)


@pytest.mark.parametrize('kind', ['EKF', 'UKF'])
def test_filters_on_a_model_written_as_expressions_equal_the_zoo_filter(kind):
    """A model defined with `set_dynamical_equations` / `set_measurement_equations` (compiled at setup, csrc/hilo_jit.hip) in the
    Kalman filters: the emitted functor is the zoo functor statement by statement; three filter steps (predict + update each)
    agree with the precompiled filter to round-off (the two translation units are free to contract multiply-adds differently)."""
    import hilo_mpc_amd as H
    from tests.problems import symbolic_model
    x, P, u, p, y = _chemo_batch(200, seed=3)
    out = []
    for m in (H.Model('chemostat4'), symbolic_model('chemostat4')):
        f = getattr(H, kind)(m.discretize('erk', order=4).setup(dt=1.))
        f.setup()
        f.Q, f.R = 1e-4, 1e-2
        f.set_initial_guess(x, P0=P)
        for _ in range(3):
            sol = f.estimate(y=y, u=u, p=p)
        out.append((f.x.cpu().numpy().copy(), f.P.cpu().numpy().copy(), np.asarray(sol['y']).copy()))
    for a, b in zip(*out):
        np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-14)
    # a model without a zoo twin: the DAE pendulum (algebraic state eliminated inside the functor), one EKF step stays finite
    md = symbolic_model('pendulum4_dae').discretize('rk4').setup(dt=.05)
    f = H.EKF(md)
    f.setup()
    f.Q, f.R = 1e-4, 1e-2
    f.set_initial_guess(np.array([[2.5, 0., .1, 0.]]), P0=np.eye(4)[None])
    sol = f.estimate(y=np.array([[2.5, 0., .1, 0.]]), u=np.array([[.2]]))
    assert np.all(np.isfinite(f.x.cpu().numpy())) and np.all(np.isfinite(f.P.cpu().numpy()))


@pytest.mark.parametrize('kind', ['EKF', 'UKF'])
@pytest.mark.parametrize('symbolic', [False, True])
def test_several_steps_in_one_launch_equal_the_single_steps(kind, symbolic):
    """`estimate(steps=K)` = `self._function.mapaccum(steps)` of the reference (kf.py:296-306): one launch for K filter steps with
    per-step measurements and inputs (hilo_kf_steps), the tile of every step returned - against K calls of the single step."""
    import hilo_mpc_amd as H
    from tests.problems import symbolic_model
    K, B = 5, 300
    x, P, u, p, y = _chemo_batch(B, seed=7)
    rng = np.random.default_rng(8)
    ys = y[None] + .01 * rng.normal(size=(K, B, 2))
    us = u[None] * (1 + .1 * rng.uniform(-1, 1, (K, B, 2)))
    model = (symbolic_model('chemostat4') if symbolic else H.Model('chemostat4')).discretize('erk', order=4).setup(dt=1.)
    one, many, held = (getattr(H, kind)(model) for _ in range(3))
    for f in (one, many, held):
        f.setup()
        f.Q, f.R = 1e-4, 1e-2
        f.set_initial_guess(x, P0=P)
    xs, Ps, yps = [], [], []
    for k in range(K):
        sol = one.estimate(y=ys[k], u=us[k], p=p)
        xs.append(one.x.cpu().numpy()), Ps.append(one.P.cpu().numpy()), yps.append(np.asarray(sol['y']))
    sol = many.estimate(y=ys, u=us, p=p, steps=K)
    assert sol['x'].shape == (K, B, 4) and sol['P'].shape == (K, B, 4, 4) and sol['y'].shape == (K, B, 2)
    np.testing.assert_allclose(sol['x'], np.stack(xs), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(sol['P'], np.stack(Ps), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(sol['y'], np.stack(yps), rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(many.x.cpu().numpy(), xs[-1], rtol=1e-13, atol=1e-15)
    # inputs held over the steps; and a single step after a multi-step call continues from its last tile
    solh = held.estimate(y=ys, u=us[0], p=p, steps=K)
    ref = getattr(H, kind)(model)
    ref.setup()
    ref.Q, ref.R = 1e-4, 1e-2
    ref.set_initial_guess(x, P0=P)
    for k in range(K):
        ref.estimate(y=ys[k], u=us[0], p=p)
    np.testing.assert_allclose(solh['x'][-1], ref.x.cpu().numpy(), rtol=1e-13, atol=1e-15)
    held.estimate(y=ys[0], u=us[0], p=p)
    ref.estimate(y=ys[0], u=us[0], p=p)
    np.testing.assert_allclose(held.x.cpu().numpy(), ref.x.cpu().numpy(), rtol=1e-13, atol=1e-15)


# ---- the BASELINE batch: one instance on a team of lanes (csrc/hilo_kf_kernel.h::kf_team_body) ---------------------------------
@pytest.mark.parametrize('kind', ['EKF', 'UKF'])
def test_filters_at_the_baseline_batch_vs_oracle(kind):
    """B = 4096 instances, 16 filter steps in one launch (BASELINE config 3 as bench.py runs it) against 16 steps of the oracle.
    At this batch the library puts one instance on a team of lanes (Jacobian columns / sigma points per lane, LDS-staged).
    Tolerance: the single step's (1e-11 / 1e-10) grown over 16 steps (worst of 16384 state entries: 3e-9); UKF with alpha = 1 (see above)."""
    import hilo_mpc_amd as H
    B, K = 4096, 16
    x, P, u, p, y = _chemo_batch(B, seed=11)
    rng = np.random.default_rng(12)
    ys = y[None] + .01 * rng.normal(size=(K, B, 2))
    om = omodels.get('chemostat4').discretize(4)
    tile, yps = okf.pack(x, P), []
    for k in range(K):
        if kind == 'EKF':
            tile, yp = okf.kf_step(om, tile, ys[k], u, p, 1e-4, 1e-2, 1.)
        else:
            tile, yp = okf.ukf_step(om, tile, ys[k], u, p, 1e-4, 1e-2, 1., alpha=1.)
        yps.append(yp)
    model = H.Model('chemostat4').discretize('rk4').setup(dt=1.)
    f = H.EKF(model) if kind == 'EKF' else H.UKF(model, alpha=1.)
    f.setup()
    f.Q, f.R = 1e-4, 1e-2
    f.set_initial_guess(x, P0=P)
    sol = f.estimate(y=ys, u=u, p=p, steps=K)
    np.testing.assert_allclose(f.x.cpu().numpy(), tile[:, :, 0], rtol=2e-8, atol=1e-12)
    np.testing.assert_allclose(f.P.cpu().numpy(), tile[:, :, 1:], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(sol['y'], np.stack(yps), rtol=2e-8, atol=1e-12)
    # ... and a single fused step at this batch (the same team kernel, one step)
    g = H.EKF(model) if kind == 'EKF' else H.UKF(model, alpha=1.)
    g.setup()
    g.Q, g.R = 1e-4, 1e-2
    g.set_initial_guess(x, P0=P)
    for k in range(2):
        g.estimate(y=ys[k], u=u, p=p)
    np.testing.assert_allclose(g.x.cpu().numpy(), sol['x'][1], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(g.P.cpu().numpy(), sol['P'][1], rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize('kind,symbolic', [('EKF', False), ('UKF', False), ('EKF', True), ('UKF', True)])
def test_team_and_one_lane_kernels_agree(kind, symbolic):
    """The same instances inside a small batch (team of lanes per instance) and inside a batch large enough for one instance per
    lane (> 2 waves per SIMD of teams): same arithmetic per entry, agreement to rounding (UKF with the default alpha = 1e-3:
    the accumulation order of the weighted sums is the same in both, kf.py:535-548)."""
    import hilo_mpc_amd as H
    from tests.problems import symbolic_model
    K, small, big = 3, 777, 40000
    x, P, u, p, y = _chemo_batch(big, seed=21)
    rng = np.random.default_rng(22)
    ys = y[None] + .01 * rng.normal(size=(K, big, 2))
    model = (symbolic_model('chemostat4') if symbolic else H.Model('chemostat4')).discretize('rk4').setup(dt=1.)
    out = []
    for n in (small, big):
        f = getattr(H, kind)(model)
        f.setup()
        f.Q, f.R = 1e-4, 1e-2
        f.set_initial_guess(x[:n], P0=P[:n])
        sol = f.estimate(y=ys[:, :n], u=u[:n], p=p[:n], steps=K)
        out.append((sol['x'][:, :small], sol['P'][:, :small], sol['y'][:, :small]))
    # UKF: every kernel propagates its sigma points through csrc/hilo_kf_kernel.h::ukf_propagate - the same operations, but the
    # compiler contracts multiply-adds differently in the two kernels, and alpha = 1e-3 weighs the sigma points with ~1e6
    tol = dict(rtol=1e-12, atol=1e-14) if kind == 'EKF' else dict(rtol=1e-7, atol=1e-7)
    for a, b in zip(*out):
        np.testing.assert_allclose(a, b, **tol)


@pytest.mark.parametrize('kind,B', [('EKF', 33000), ('UKF', 8300)])
def test_one_lane_kernels_of_the_common_recipe_vs_oracle(kind, B):
    """Batches beyond two waves per SIMD of teams run one instance per lane; `discretize('rk4')` with shared Q, R takes the LEAN
    instantiations (`ekf_multi_lean_kernel`, `kf_multi_kernel<.., true, true>`): 6 filter steps in one launch against 6 steps of the
    oracle (UKF with alpha = 1 like the other oracle comparisons; the tolerance is that of the team kernels' test)."""
    import hilo_mpc_amd as H
    K = 6
    x, P, u, p, y = _chemo_batch(B, seed=41)
    rng = np.random.default_rng(42)
    ys = y[None] + .01 * rng.normal(size=(K, B, 2))
    om = omodels.get('chemostat4').discretize(4)
    tile = okf.pack(x, P)
    for k in range(K):
        if kind == 'EKF':
            tile, yp = okf.kf_step(om, tile, ys[k], u, p, 1e-4, 1e-2, 1.)
        else:
            tile, yp = okf.ukf_step(om, tile, ys[k], u, p, 1e-4, 1e-2, 1., alpha=1.)
    model = H.Model('chemostat4').discretize('rk4').setup(dt=1.)
    f = H.EKF(model) if kind == 'EKF' else H.UKF(model, alpha=1.)
    f.setup()
    f.Q, f.R = 1e-4, 1e-2
    f.set_initial_guess(x, P0=P)
    sol = f.estimate(y=ys, u=u, p=p, steps=K)
    np.testing.assert_allclose(f.x.cpu().numpy(), tile[:, :, 0], rtol=2e-8, atol=1e-12)
    np.testing.assert_allclose(f.P.cpu().numpy(), tile[:, :, 1:], rtol=1e-7, atol=1e-12)
    np.testing.assert_allclose(sol['y'][-1], yp, rtol=2e-8, atol=1e-12)


@pytest.mark.parametrize('kind,B', [('EKF', 40000), ('UKF', 40000), ('UKF', 140000)])
def test_lean_variant_of_the_multi_step_kernel_equals_the_general_one(kind, B):
    """`discretize('rk4')` with Q, R shared by the batch runs `kf_multi_kernel<.., LEAN>` / `ekf_multi_lean_kernel` (one Runge-Kutta
    slope alive, Q and R read where they are used: two waves per SIMD); the same values handed over per instance ([B, n, n]) take
    the general kernel.  EKF: same terms in the same order, equal to rounding of the fused multiply-adds the compiler forms."""
    import hilo_mpc_amd as H
    import torch
    K = 4       # (UKF from 2 x 1024 x 64 instances on: the instantiation with one sigma point at a time, two waves per SIMD)
    x, P, u, p, y = _chemo_batch(B, seed=31)
    rng = np.random.default_rng(32)
    ys = y[None] + .01 * rng.normal(size=(K, B, 2))
    model = H.Model('chemostat4').discretize('rk4').setup(dt=1.)
    Q, R = np.diag([1e-4, 2e-4, 3e-4, 1e-4]) + 1e-5, np.array([[1e-2, 1e-3], [1e-3, 2e-2]])
    out = []
    for per_instance in (False, True):
        f = getattr(H, kind)(model)
        f.setup()
        if per_instance:
            f.Q = torch.as_tensor(np.tile(Q, (B, 1, 1)), device='cuda')
            f.R = torch.as_tensor(np.tile(R, (B, 1, 1)), device='cuda')
        else:
            f.Q, f.R = Q, R
        f.set_initial_guess(x, P0=P)
        sol = f.estimate(y=ys, u=u, p=p, steps=K)
        out.append((sol['x'], sol['P'], sol['y']))
    tol = dict(rtol=1e-13, atol=1e-15) if kind == 'EKF' else dict(rtol=1e-7, atol=1e-7)       # (UKF: see the team kernels' test)
    for a, b in zip(*out):
        np.testing.assert_allclose(a, b, **tol)

