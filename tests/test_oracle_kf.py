"""Pin oracle/kf.py against the reference's own known-answer tests (tests/test_KFs.py in the reference)."""
import numpy as np

from oracle import kf, models


def _lin():
    return models.get('linear2').discretize(order=1)     # test_KFs.py:254 `discretize('erk', order=1)`


def test_kf_predict_kat():
    # reference tests/test_KFs.py:488-503
    m = _lin()
    xP = np.array([[.8, 1., 0.], [0., 0., 1.]])
    out = kf.kf_predict(m, xP, [.8], [.5, .4], .01 * np.eye(2), 1.)
    np.testing.assert_allclose(out[0], np.array([[1.2, .26, .25], [.4, .25, .62]]))


def test_kf_update_kat():
    # reference tests/test_KFs.py:505-522
    m = _lin()
    pred = np.array([[1.2, .26, .25], [.4, .25, .62]])
    up, yp = kf.kf_update(m, pred, [.322052], [.8], [.5, .4], .064, 1.)
    np.testing.assert_allclose(up[0], np.array([[1.17151023, .16862573, .023391813],
                                                [.32934538, .023391813, .0580117]]), rtol=1e-7)
    np.testing.assert_allclose(yp, np.array([[.4]]))


def test_kf_one_step_kat():
    # reference tests/test_KFs.py:298-316
    m = _lin()
    xP = kf.pack([.8, 0.], np.eye(2))
    out, _ = kf.kf_step(m, xP, [.3894626], [.8], [.5, .4], [.01, .01], .064, 1.)
    np.testing.assert_allclose(out[0, :, 0], np.array([1.19614861, .39044856]), rtol=1e-7)


def test_ekf_one_step_kat():
    # reference tests/test_KFs.py:548-566
    m = models.get('toy1d')
    out, _ = kf.kf_step(m, kf.pack([9.], np.eye(1)), [2.59109], np.zeros((1, 0)), np.zeros((1, 0)), 10., 1., 1.)
    np.testing.assert_allclose(out[0, :, 0], np.array([7.206059]), rtol=1e-7)


def test_ukf_one_step_kat():
    # reference tests/test_KFs.py:606-624
    m = models.get('toy1d')
    out, _ = kf.ukf_step(m, kf.pack([9.], np.eye(1)), [2.59109], np.zeros((1, 0)), np.zeros((1, 0)), 10., 1., 1.)
    np.testing.assert_allclose(out[0, :, 0], np.array([7.2739647]), rtol=1e-7)


P_BIO = [.15, 303.15, .13, .00025, 15., .14]


def test_ukf_sigma_predict_kat():
    # reference tests/test_KFs.py:716-734 (alpha=1, continuous model, dt=.1, atol 1e-3 as in the reference)
    m = models.get('bioreactor3')
    xP = kf.pack([299.876, .217108, 20.], np.eye(3))
    out = kf.ukf_predict(m, xP, [.01], P_BIO, np.zeros((3, 3)), .1, alpha=1.)
    ref = np.array([[299.925, 0.970453, 1.08254e-06, -2.11206e-05, 299.925, 301.631, 299.925, 299.925, 298.218,
                     299.925, 299.925],
                    [0.219433, 1.08254e-06, 1.02165, -0.0848792, 0.219446, 0.219451, 1.97011, 0.219537, 0.21944,
                     -1.53128, 0.219345],
                    [19.9619, -2.11206e-05, -0.0848792, 1.00427, 19.9618, 19.9617, 19.8164, 21.6914, 19.9618,
                     20.1075, 18.2322]])
    np.testing.assert_allclose(out[0], ref, atol=1e-3)


def test_ukf_sigma_update_kat():
    # reference tests/test_KFs.py:736-756
    m = models.get('bioreactor3')
    x = np.array([[299.925], [.219433], [19.9619]])
    P = np.array([[.970453, 1.08254e-06, -2.11206e-05], [1.08254e-06, 1.02165, -.0848792],
                  [-2.11206e-05, -.0848792, 1.00427]])
    X = np.array([[299.925, 301.631, 299.925, 299.925, 298.218, 299.925, 299.925],
                  [.219446, .219451, 1.97011, .219537, .21944, -1.53128, .219345],
                  [19.9618, 19.9617, 19.8164, 21.6914, 19.9618, 20.1075, 18.2322]])
    up, yp = kf.ukf_update(m, np.concatenate([x, P, X], axis=1), [300.941, .245805], [.01], P_BIO,
                           np.diag([.25, .01]), .1, alpha=1.)
    np.testing.assert_allclose(up[0], np.array([[300.733, 0.198539, -2.03814e-06, 1.56415e-06],
                                                [0.245549, -2.03814e-06, 0.00990874, -0.000819447],
                                                [19.9597, 1.56415e-06, -0.000819447, 0.997286]]), atol=1e-3)
    np.testing.assert_allclose(yp[0], np.array([299.925, .219434]), atol=1e-3)


def test_ukf_weights_defaults():
    g, W = kf.ukf_weights(4)
    lam = 1e-6 * 4 - 4
    assert np.isclose(g, np.sqrt(4 + lam))
    assert np.isclose(W[0].sum(), 1.0)


def test_batched_matches_loop():
    rng = np.random.default_rng(0)
    m = models.get('chemostat4').discretize(4)
    B = 5
    x = np.array([.1, 40., 0.5, 0.2]) * (1 + .1 * rng.uniform(-1, 1, (B, 4)))
    A = rng.normal(size=(B, 4, 4)) * .1
    P = A @ np.swapaxes(A, 1, 2) + np.eye(4) * .5
    u = rng.uniform(0, .3, (B, 2))
    p = np.tile([100., 4., 1., 0.], (B, 1))
    y = x[:, [0, 2]] + .01 * rng.normal(size=(B, 2))
    for step in (kf.kf_step, kf.ukf_step):
        full, _ = step(m, kf.pack(x, P), y, u, p, 1e-4, 1e-2, 1.)
        for b in range(B):
            one, _ = step(m, kf.pack(x[b], P[b]), y[b], u[b], p[b], 1e-4, 1e-2, 1.)
            np.testing.assert_allclose(full[b], one[0], rtol=1e-12, atol=1e-14)
