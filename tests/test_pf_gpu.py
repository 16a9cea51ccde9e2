"""GPU parity of the particle filter: the function the reference builds at setup() - propagate, measure, weigh -
the resampling and the statistics of the particle set on the device against the numpy restatement (oracle/pf.py), and whole
seeded runs of `estimate()` draw by draw (the random numbers come from numpy's global generator in the reference's order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from hilo_mpc_amd import Model, PF                     # noqa: E402
from oracle import models as omodels, pf as opf        # noqa: E402


def _pend1():
    """A model written as expressions with ONE measurement (the reference's likelihood case): the filter kernels are compiled
    at setup()."""
    import sympy as sp
    from hilo_mpc_amd.expr import sin
    from oracle.models import OracleModel
    m = Model(name='pend1')
    x, u = m.set_dynamical_states(['th', 'om']), m.set_inputs(['tau'])
    m.set_dynamical_equations([x[1], -sin(x[0]) - .1 * x[1] + u[0]])
    m.set_measurement_equations([x[0]])
    th, w, tau = sp.symbols('th om tau')
    om = OracleModel('pend1', -1, [th, w], [tau], [], [w, -sp.sin(th) - .1 * w + tau], [th])
    return m.discretize('erk', order=4).setup(dt=.1), om.discretize(4), .1


def _case(name):
    if name == 'pend1':
        return _pend1()
    m = Model(name)
    m = m.setup(dt=1.) if name == 'toy1d' else m.discretize('rk4').setup(dt=.5)
    om = omodels.get(name)
    return m, (om if om.discrete else om.discretize(4)), m.dt


POINT = {'toy1d': ([1.5], [], []), 'chemostat4': ([.1, 40., .5, .2], [.1, .2], [100., 4., 1., 0.]), 'pend1': ([.5, 0.], [.2], [])}


@pytest.mark.parametrize('name', ['toy1d', 'chemostat4', 'pend1'])
def test_function_resampling_and_statistics_vs_oracle(name):
    m, om, dt = _case(name)
    x0, u, p = POINT[name]
    nx, ny = m.n_x, m.n_y
    pf = PF(m)
    pf.setup(n_samples=500)
    rng = np.random.default_rng(3)
    B, N = 3, 500
    X = np.asarray(x0) * (1 + .1 * rng.standard_normal((B, N, nx)))
    w = .01 * np.abs(np.asarray(x0)) * rng.standard_normal((B, N, nx))
    R = np.diag(rng.uniform(.5, 2., ny)) * 1e-2 * (1. if name != 'chemostat4' else 1e-2)
    v = rng.standard_normal((B, N, ny)) @ np.sqrt(R)
    xn = om.f(np.asarray(x0)[None], np.asarray(u)[None], np.asarray(p)[None], dt)
    yref = om.h(xn, np.asarray(u)[None], np.asarray(p)[None], dt)[0]          # measurement of the propagated nominal state
    y = yref * (1 + .02 * rng.standard_normal((B, ny)))
    up = np.concatenate([u, p])[None] if (len(u) + len(p)) else None
    Xp, Y, q = pf.function(X, y, up, w, v, R=R)
    for b in range(B):
        Xr, Yr, qr = opf.pf_function(om, dt, X[b], y[b], u, p, w[b], v[b], R)
        np.testing.assert_allclose(Xp[b].cpu().numpy(), Xr, rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(Y[b].cpu().numpy(), Yr, rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(q[b].cpu().numpy(), qr, rtol=1e-9, atol=1e-300)
        assert abs(q[b].sum().item() - 1.) < 1e-12
    # resampling: numpy's choice algorithm with the same uniforms
    import torch
    from hilo_mpc_amd import _lib
    from hilo_mpc_amd._device import ptr, stream_ptr, to_dev
    uni = rng.random((B, N))
    Xs, Ys = torch.empty_like(Xp), torch.empty_like(Y)
    ind = torch.empty(B, N, dtype=torch.int32, device=Xp.device)
    _lib.check(_lib.lib().hilo_pf_resample(pf._handle, B, N, ptr(Xp), ptr(Y), ptr(q), ptr(to_dev(uni, Xp.device)), ptr(Xs), ptr(Ys),
                                           ptr(ind), stream_ptr(Xp.device)))
    for b in range(B):
        qb = q[b].cpu().numpy()
        cdf = qb.cumsum()
        cdf /= cdf[-1]
        ref = cdf.searchsorted(uni[b], side='right')
        got = ind[b].cpu().numpy()
        assert np.mean(got == ref) == 1. or (np.mean(got == ref) > .995 and np.all(np.abs(got - ref) <= 1))
        np.testing.assert_array_equal(Xs[b].cpu().numpy(), Xp[b].cpu().numpy()[got])
        np.testing.assert_array_equal(Ys[b].cpu().numpy(), Y[b].cpu().numpy()[got])
    # statistics (with a roughening increment)
    add = 1e-3 * rng.standard_normal((B, N, nx))
    Xa = Xs.clone()
    xm, ym, P, lo, hi = pf._stats(Xa, Ys, to_dev(add, Xp.device))
    Xh = Xs.cpu().numpy() + add
    np.testing.assert_allclose(Xa.cpu().numpy(), Xh, rtol=0, atol=1e-15)
    for b in range(B):
        np.testing.assert_allclose(xm[b].cpu().numpy(), Xh[b].mean(axis=0), rtol=1e-12)
        np.testing.assert_allclose(ym[b].cpu().numpy(), Ys[b].cpu().numpy().mean(axis=0), rtol=1e-12)
        np.testing.assert_allclose(P[b].cpu().numpy(), np.atleast_2d(np.cov(Xh[b].T)), rtol=1e-9, atol=1e-18)
        np.testing.assert_array_equal(lo[b].cpu().numpy(), Xh[b].min(axis=0))
        np.testing.assert_array_equal(hi[b].cpu().numpy(), Xh[b].max(axis=0))


@pytest.mark.parametrize('name,roughening,prior', [('toy1d', False, False), ('toy1d', True, False), ('pend1', True, True)])
def test_seeded_run_equals_the_oracle_draw_by_draw(name, roughening, prior):
    """`estimate()` over several steps with numpy's generator seeded: the same particles, indices, mean and covariance as the
    restated reference flow (initial sample, process / measurement noise, prior editing, resampling, roughening)."""
    m, om, dt = _case(name)
    x0, u, _ = POINT[name]
    nx, ny, N = m.n_x, m.n_y, 60
    Q, R = np.eye(nx) * (1. if name == 'toy1d' else 1e-4), np.eye(ny) * (1. if name == 'toy1d' else 1e-3)
    rng = np.random.default_rng(8)
    ys = []
    xt = np.asarray(x0, dtype=float)
    for _ in range(5):
        xt = om.f(xt[None], np.asarray(u)[None], np.zeros((1, 0)), dt)[0]
        ys.append(om.h(xt[None], np.asarray(u)[None], np.zeros((1, 0)), dt)[0] + np.sqrt(np.diag(R)) * rng.standard_normal(ny))
    pf = PF(m, roughening=roughening, prior_editing=prior)
    pf.setup(n_samples=N)
    pf.Q, pf.R = Q, R
    pf.set_initial_guess(x0, P0=np.eye(nx) * (2. if name == 'toy1d' else 1e-2))
    ref = opf.ParticleFilter(om, dt, n_samples=N, roughening=roughening, prior_editing=prior)
    ref.Q, ref.R = Q, R
    ref.set_initial_guess(x0, np.eye(nx) * (2. if name == 'toy1d' else 1e-2))
    np.random.seed(21)
    got = []
    for y in ys:
        s = pf.estimate(y=y, u=u if len(u) else None)
        got.append((np.array(s['x']), np.array(s['P']), np.array(s['X']), pf._last['index'].cpu().numpy()[0]))
    np.random.seed(21)
    for k, y in enumerate(ys):
        r = ref.estimate(y, u)
        x, P, X, ind = got[k]
        assert np.array_equal(ind, r['index']), k
        np.testing.assert_allclose(X[0].T, r['X'], rtol=1e-8, atol=1e-10)          # (the benchmark map amplifies round-off)
        np.testing.assert_allclose(x[:, 0], r['x'], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(P[0], r['P'], rtol=1e-7, atol=1e-12)
    assert x.shape == (nx, 1)


def test_batch_of_filters_tracks_the_state():
    """256 filters x 512 particles on the reference's benchmark system: estimates stay with the true trajectories (the
    measurement x^2/20 leaves the sign open - compared in |x|)."""
    m, om, dt = _case('toy1d')
    B, N = 256, 512
    rng = np.random.default_rng(1)
    xt = rng.uniform(2., 6., (B, 1))
    pf = PF(m, roughening=True)
    pf.setup(n_samples=N)
    pf.Q, pf.R = [[.5]], [[.1]]
    pf.set_initial_guess(np.abs(xt) + rng.standard_normal((B, 1)), P0=[[2.]])
    pf.probability_density_function = lambda mu, s, n: np.random.multivariate_normal(mu, s, size=n)
    np.random.seed(2)
    err = []
    for _ in range(6):
        xt = om.f(xt, np.zeros((B, 0)), np.zeros((B, 0)), dt) + np.sqrt(.5) * rng.standard_normal((B, 1))
        y = om.h(xt, np.zeros((B, 0)), np.zeros((B, 0)), dt) + np.sqrt(.1) * rng.standard_normal((B, 1))
        s = pf.estimate(y=y)
        err.append(np.abs(np.abs(np.asarray(s['x'])) - np.abs(xt)))
    assert np.median(err[-1]) < 1.5 and np.all(np.isfinite(np.asarray(s['P'])))
