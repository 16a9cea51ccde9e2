"""GPU parity: kernels, means, exact GP inference/prediction vs the reference's known answers and the oracle."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import gp as ogp                                  # noqa: E402
from tests.util import kernel_from_spec, mean_from_spec       # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
KCASES = json.load(open(os.path.join(GOLD, 'kernels_kat.json')))
MCASES = json.load(open(os.path.join(GOLD, 'means_kat.json')))


@pytest.mark.parametrize('case', KCASES, ids=[f"{c['ref_test'].split('::')[-1]}@{c['ref_line']}" for c in KCASES])
def test_kernel_kat(case):
    k = kernel_from_spec(case['spec'])
    K = k(*[np.array(a) for a in case['args']])
    assert isinstance(K, np.ndarray)
    np.testing.assert_allclose(K, np.array(case['expected']), rtol=case['tol'].get('rtol', 1e-7),
                               atol=case['tol'].get('atol', 0))
    # and against the oracle at fp64 tolerance (transcendentals differ by <= a few ulp between libm and ocml)
    np.testing.assert_allclose(K, ogp.kernel(case['spec'], *[np.array(a) for a in case['args']]), rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize('case', MCASES, ids=[f"{c['ref_test'].split('::')[-1]}@{c['ref_line']}" for c in MCASES])
def test_mean_kat(case):
    mu = mean_from_spec(case['spec'])(*[np.array(a) for a in case['args']])
    np.testing.assert_allclose(mu, np.array(case['expected']), rtol=case['tol'].get('rtol', 1e-7),
                               atol=case['tol'].get('atol', 0))


def test_gp_lml_kat():
    from hilo_mpc_amd import GP
    gp = GP(['x', 'y'], 'z')                                              # test_GPs.py:325-329, 354-363
    X = np.array([[0., .5, 1. / np.sqrt(2.), np.sqrt(3.) / 2., 1., 0.],
                  [1., np.sqrt(3.) / 2., 1. / np.sqrt(2.), .5, 0., -1.]])
    y = np.array([[0., np.pi / 6., np.pi / 4., np.pi / 3., np.pi / 2., np.pi]])
    with pytest.raises(RuntimeError, match="The training data has not been set"):
        gp.setup()
    gp.set_training_data(X, y)
    gp.setup()
    np.testing.assert_approx_equal(gp.log_marginal_likelihood(), -9.82229944)


def test_gp_rasmussen_kat_and_predict_properties():
    from hilo_mpc_amd import GP, Kernel, Mean
    kernel = Kernel.matern_32(length_scales=.25)                          # test_GPs.py:1102-1123
    mean = Mean.linear(coefficient=.5) + Mean.one()
    x = ogp.park_miller_randn(.3, (20, 1))
    K = kernel(x.T)
    mu = mean(x.T)
    y = np.linalg.cholesky(K) @ ogp.park_miller_randn(.15, (20, 1)) + mu.T + .1 * ogp.park_miller_randn(.2, (20, 1))
    gp = GP(['x'], ['y'], mean=mean, kernel=kernel, noise_variance=.1 ** 2)
    gp.set_training_data(x.T, y.T)
    gp.setup()
    np.testing.assert_approx_equal(gp.log_marginal_likelihood(), -11.9706317)
    xs = np.linspace(-1.9, 1.9, 101).reshape(1, -1)
    m1, v1 = gp.predict(xs)                                               # test_GPs.py:654-666
    m2, v2 = gp.predict(xs, noise_free=True)
    np.testing.assert_allclose(m1, m2)
    assert np.all(v2 < v1)
    ker = {'type': 'matern_32', 'kwargs': {'length_scales': .25}}
    mus = {'type': 'sum', 'children': [{'type': 'linear', 'kwargs': {'coefficient': .5}}, {'type': 'one'}]}
    post = ogp.Posterior(ker, mus, x.T, y.T, .1 ** 2)
    mo, vo = post.predict(xs)
    np.testing.assert_allclose(m1, mo, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(v1, vo, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('n,m', [(200, 2048), (37, 5), (1, 1)])
def test_gp_predict_vs_oracle_config_c4(n, m):
    """SE-ARD l = [10, 1], sf2 = 1, sn2 = 1e-4, features (S, I), 200 training points on a grid."""
    import torch
    from hilo_mpc_amd import GP, Kernel
    rng = np.random.default_rng(20260926)
    if n == 200:
        S, I = np.meshgrid(np.linspace(0, 40, 20), np.linspace(0, 4, 10))
        X = np.stack([S.ravel(), I.ravel()])
    else:
        X = np.stack([rng.uniform(0, 40, n), rng.uniform(0, 4, n)])
    phi = .407 * X[0] / (.108 + X[0] + X[0] ** 2 / 14814.)
    y = (phi * (1. + .22 * 0. / (.22 + X[1])) + 1e-2 * rng.normal(size=X.shape[1]))[None, :]
    gp = GP(['S', 'I'], 'mu', kernel=Kernel.squared_exponential(active_dims=[0, 1], length_scales=[10., 1.], ard=True),
            noise_variance=1e-4)
    gp.set_training_data(X, y)
    gp.setup()
    spec = {'type': 'squared_exponential', 'kwargs': {'active_dims': [0, 1], 'length_scales': [10., 1.], 'ard': True}}
    post = ogp.Posterior(spec, {'type': 'zero'}, X, y, 1e-4)
    # conditioning of K + sn2 I is ~1e6..1e7 here: alpha (and hence the mean) agree to ~cond * eps
    np.testing.assert_allclose(gp.log_marginal_likelihood(), post.lml, rtol=1e-9)
    Xq = np.stack([rng.uniform(0, 40, m), rng.uniform(0, 4, m)])
    mo, vo = post.predict(Xq)
    mg, vg = gp.predict(Xq)
    np.testing.assert_allclose(mg, mo, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(vg, vo, rtol=1e-6, atol=1e-9)
    # device-resident queries give device results; mean-only path
    md, vd = gp.predict(torch.as_tensor(Xq, device='cuda'), return_var=False)
    assert vd is None and isinstance(md, torch.Tensor)
    np.testing.assert_allclose(md.cpu().numpy(), mg, rtol=0, atol=0)
    me, ve = gp.predict(np.zeros((2, 0)))
    assert me.shape == (1, 0)


def test_gp_not_positive_definite_is_an_error():
    from hilo_mpc_amd import GP, Kernel
    gp = GP(['x'], 'y', kernel=Kernel.constant(), noise_variance=0.)
    gp.set_training_data(np.array([[1., 2., 3.]]), np.array([[1., 2., 3.]]))
    with pytest.raises(ValueError, match="not positive definite"):
        gp.setup()


@pytest.mark.parametrize('make,shift,expected,rtol', [
    (lambda K: None, 0., [.0085251, .5298217, .8114553], 1e-5),                       # default kernel: squared exponential
    (lambda K: K.constant(), 3., [.7009480, 3.0498634], 1e-5),
    (lambda K: K.matern_32(), 0., [.0088262, .8329355, .9398366], 1e-5),
    (lambda K: K.neural_network(), 0., [.0095177, 5.7756069, .1554265], 1e-5),
    (lambda K: K.matern_52(), 0., [.0086694, .7180206, .9571137], 1e-5),
    (lambda K: K.piecewise_polynomial(1), 0., [.0088391, 1.6079331, .6545336], 1e-5),
    (lambda K: K.piecewise_polynomial(2), 0., [.0088244, 2.0883167, .852502], 1e-5),
    (lambda K: K.piecewise_polynomial(3), 0., [.0086766, 2.247363, .7873581], 1e-5),
    (lambda K: K.polynomial(3), 0., [.0980796, 1.3112287, .5083423], 1e-5),
    (lambda K: K.linear(), 0., [.6627861, .008198], 1e-4),
    (lambda K: K.periodic(), 0., [.4975112, .159969, .5905631, .8941061], 1e-3),
], ids=['SE', 'Const', 'M32', 'NN', 'M52', 'PP1', 'PP2', 'PP3', 'Poly', 'Lin', 'Periodic'])
def test_fit_model_reference_kats(make, shift, expected, rtol):
    """`GaussianProcess.fit_model()` on the device objective against the reference's fitted-value known answers
    (tests/test_GPs.py:846-904, data set :835-838); a successful fit does not warn (:912-916)."""
    import warnings
    from hilo_mpc_amd import GP, Kernel
    x = ogp.park_miller_randn(.8, (20, 1))
    y = np.sin(3 * x) + .1 * ogp.park_miller_randn(.9, (20, 1)) + shift
    g = GP(['x'], ['y'], kernel=make(Kernel), noise_variance=np.exp(-2))
    g.set_training_data(x.T, y.T)
    g.setup()
    lml0 = g.log_marginal_likelihood()
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        g.fit_model()
    np.testing.assert_allclose(g.hyperparameter_values, expected, rtol=max(rtol, 2e-6))
    assert g.log_marginal_likelihood() > lml0 and g._optimization_stats['success']
    xs = np.linspace(-3, 3, 61).reshape(1, -1)
    mean, var = g.predict(xs)                                                            # the fitted model is set up
    post = ogp.Posterior(_spec_of(g.kernel), {'type': 'zero'}, x.T, y.T, g.noise_variance)
    mo, vo = post.predict(xs)
    np.testing.assert_allclose(mean, mo, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(var, vo, rtol=1e-5, atol=1e-9)


def _spec_of(k):
    t = {'SE': 'squared_exponential', 'Const': 'constant', 'M32': 'matern_32', 'NN': 'neural_network', 'M52': 'matern_52',
         'PP': 'piecewise_polynomial', 'Poly': 'polynomial', 'Lin': 'linear', 'Periodic': 'periodic'}[k.acronym]
    kw = {a: getattr(k, a) for a in k._hyper}
    if hasattr(k, 'degree'):
        kw['degree'] = k.degree
    return {'type': t, 'kwargs': kw}


def test_hyperprior_kats():
    """Hyper-priors on the device objective: the reference's LML with a Laplace prior on the length scale
    (tests/test_GPs.py:366-385) and its fit under a Gaussian prior on the noise variance (:596-621); plain fit (:578-591)."""
    from hilo_mpc_amd import GP
    X = np.array([[0., .5, 1. / np.sqrt(2.), np.sqrt(3.) / 2., 1., 0.],
                  [1., np.sqrt(3.) / 2., 1. / np.sqrt(2.), .5, 0., -1.]])
    y = np.array([[0., np.pi / 6., np.pi / 4., np.pi / 3., np.pi / 2., np.pi]])
    g = GP(['x', 'y'], 'z')
    g.set_training_data(X, y)
    g.setup()
    g.set_hyperprior('SE.length_scales', 'Laplace', mean=0., variance=1.)
    np.testing.assert_approx_equal(g.log_marginal_likelihood(), -10.16887303)
    g.set_hyperprior('SE.length_scales', None)
    np.testing.assert_approx_equal(g.log_marginal_likelihood(), -9.82229944)
    before = g.log_marginal_likelihood()
    g.fit_model()
    assert g.log_marginal_likelihood() > before and g.noise_variance < 1e-3
    h = GP(['x', 'y'], 'z')
    h.set_hyperprior('GP.noise_variance', 'Gaussian', mean=.2, variance=.01)
    h.set_training_data(X, y + np.array([[.23757934, .55730318, .02598826, .06349002, .26647032, -.137302]]))
    h.setup()
    before = h.log_marginal_likelihood()
    h.fit_model()
    assert h.log_marginal_likelihood() > before
    np.testing.assert_allclose(h.noise_variance, 1.406995, rtol=1e-6)


def test_fit_then_predict_quantiles_like_the_reference_tests():
    """tests/test_GPs.py:625-740: fit on six noisy points, noise-free prediction has the same mean and a smaller variance,
    and the predictive quantiles bracket the mean and both generating functions."""
    from hilo_mpc_amd import GP
    X = np.array([[0., .5, 1. / np.sqrt(2.), np.sqrt(3.) / 2., 1., 0.],
                  [1., np.sqrt(3.) / 2., 1. / np.sqrt(2.), .5, 0., -1.]])
    y = np.array([[0., np.pi / 6., np.pi / 4., np.pi / 3., np.pi / 2., np.pi]])
    y = y + np.array([[0.05850223, 0.09876431, -0.05570195, 0.15573265, 0.03278181, -0.06901315]])
    g = GP(['x', 'y'], 'z')
    with pytest.raises(RuntimeError, match="has not been set up yet"):
        g.predict(np.array([0., 1.]))
    g.set_training_data(X, y)
    g.setup()
    g.fit_model()
    Xq = np.array([[.25, .6, .75, .9, .4], [.9, .75, .6, .25, -.5]])
    mean, var = g.predict(Xq)
    mean_nf, var_nf = g.predict(Xq, noise_free=True)
    np.testing.assert_allclose(mean, mean_nf)
    np.testing.assert_array_less(var_nf, var)
    y_sin = np.array([[np.arcsin(.25), np.arcsin(.6), np.arcsin(.75), np.arcsin(.9), np.pi + np.arcsin(-.4)]])
    y_cos = np.array([[np.arccos(.9), np.arccos(.75), np.arccos(.6), np.arccos(.25), np.arccos(-.5)]])
    for lb, ub in (g.predict_quantiles(X_query=Xq), g.predict_quantiles(quantiles=(1., 99.), mean=mean_nf, var=var_nf)):
        for inner in (mean, y_sin, y_cos):
            np.testing.assert_array_less(lb, inner)
            np.testing.assert_array_less(inner, ub)
    assert g.predict_quantiles() is None
    from scipy.stats import norm
    lb, ub = g.predict_quantiles(X_query=Xq)
    np.testing.assert_allclose(ub, mean_nf + norm.ppf(.975) * np.sqrt(var_nf + g.noise_variance), rtol=1e-12)


def test_device_lml_gradient_vs_oracle_trace_formula():
    """hilo_gp_lml_gradient (1/2 tr((alpha alpha^T - K^-1) dK/dtheta) on the device, one factorisation for every
    hyper-parameter) against the oracle's trace formula and against central differences of the LML itself."""
    import ctypes as C
    from hilo_mpc_amd import GP, Kernel, _lib
    from oracle import gp_fit
    x = ogp.park_miller_randn(.8, (20, 1))
    y = np.sin(3 * x) + .1 * ogp.park_miller_randn(.9, (20, 1))
    for make, otype, names in ((lambda: Kernel.squared_exponential(length_scales=.7, signal_variance=1.3), 'squared_exponential',
                                ['length_scales', 'signal_variance']),
                               (lambda: Kernel.matern_52(length_scales=.9, signal_variance=.8), 'matern_52',
                                ['length_scales', 'signal_variance'])):
        g = GP(['x'], ['y'], kernel=make(), noise_variance=.05)
        g.set_training_data(x.T, y.T)
        g.setup()
        th = np.log(np.asarray(g.hyperparameter_values))
        h = 1e-5
        progs, noise = [], []
        for i in range(th.size):
            for sgn in (1., -1.):
                e = np.zeros_like(th)
                e[i] = sgn * h
                g._set_hyperparameters(np.exp(th + e))
                progs.append(np.asarray(g.kernel.program(1), dtype=np.float64))
                noise.append(float(g.noise_variance))
        g._set_hyperparameters(np.exp(th))
        progs, noise, hh = np.ascontiguousarray(np.stack(progs)), np.array(noise), np.full(th.size, h)
        out = np.zeros(th.size)
        _lib.check(_lib.lib().hilo_gp_lml_gradient(g._handle, th.size, progs.ctypes.data, noise.ctypes.data, hh.ctypes.data,
                                                   out.ctypes.data))
        ref = gp_fit.lml_gradient(otype, names, th, x.T, y.T)
        np.testing.assert_allclose(out, ref, rtol=1e-7, atol=1e-9)
        fd = np.zeros_like(th)
        for i in range(th.size):
            e = np.zeros_like(th)
            e[i] = 1e-6
            fd[i] = -(gp_fit.negative_lml(otype, names, th + e, x.T, y.T) - gp_fit.negative_lml(otype, names, th - e, x.T, y.T)) / 2e-6
        np.testing.assert_allclose(out, fd, rtol=2e-5, atol=1e-7)


def test_fit_model_bounds_fixed_and_boxed():
    """`bounds=` of the kernel factories (util/machine_learning.py:283-334): 'fixed' keeps a hyper-parameter at its value, a
    pair boxes it; and an indefinite trial point inside the optimiser never leaves the object inconsistent."""
    from hilo_mpc_amd import GP, Kernel
    x = ogp.park_miller_randn(.8, (20, 1))
    y = np.sin(3 * x) + .1 * ogp.park_miller_randn(.9, (20, 1))
    free = GP(['x'], ['y'], kernel=Kernel.squared_exponential(), noise_variance=np.exp(-2))
    free.set_training_data(x.T, y.T)
    free.setup()
    free.fit_model()
    g = GP(['x'], ['y'], kernel=Kernel.squared_exponential(signal_variance=.5, bounds={'signal_variance': 'fixed'}),
           noise_variance=np.exp(-2))
    g.set_training_data(x.T, y.T)
    g.setup()
    g.fit_model()
    nv, ls, sv = g.hyperparameter_values
    assert sv == .5 and abs(ls - free.hyperparameter_values[1]) > 1e-3 and g._optimization_stats['success']
    from oracle import gp_fit
    ref, _ = gp_fit.fit('squared_exponential', ['length_scales'], x.T, y.T, noise_variance=np.exp(-2), fixed={'signal_variance': .5})
    np.testing.assert_allclose([nv, ls], ref, rtol=2e-5)
    b = GP(['x'], ['y'], kernel=Kernel.squared_exponential(bounds={'length_scales': (.7, 5.)}), noise_variance=np.exp(-2))
    b.set_training_data(x.T, y.T)
    b.setup()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        b.fit_model()
    assert abs(b.hyperparameter_values[1] - .7) < 1e-9                  # the free optimum 0.53 lies below the box
    mean, var = b.predict(np.linspace(-2, 2, 9).reshape(1, -1))         # consistent, factorised object
    assert np.all(np.isfinite(mean)) and np.all(var > 0)


def test_fit_model_fits_the_mean_hyperparameters_too():
    """gp.py:408-414 on the device: the bias of a constant mean is an optimisation variable next to the kernel's; at the optimum
    it equals the generalised-least-squares value (1^T Ky^-1 y) / (1^T Ky^-1 1) of the fitted covariance."""
    import warnings
    from hilo_mpc_amd import GP, Kernel, Mean
    rng = np.random.default_rng(4)
    X = np.linspace(0, 6, 25)[None]
    y = (2.5 + np.sin(X) + .05 * rng.standard_normal(X.shape))
    g = GP(['x'], ['y'], kernel=Kernel.squared_exponential(), mean=Mean.constant(.2), noise_variance=.05)
    g.set_training_data(X, y)
    g.setup()
    l0 = g.log_marginal_likelihood()
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        g.fit_model()
    assert g.log_marginal_likelihood() > l0 + 1.
    K = np.asarray(g.kernel(X, X)) + g.noise_variance * np.eye(25)
    one = np.ones(25)
    b_star = (one @ np.linalg.solve(K, y.ravel())) / (one @ np.linalg.solve(K, one))
    np.testing.assert_allclose(g.mean.bias, b_star, rtol=1e-4)
    mean, _ = g.predict(np.array([[100.]]))                              # far from the data the prediction is the mean function
    np.testing.assert_allclose(np.asarray(mean).ravel()[0], g.mean.bias, rtol=1e-6)
    held = GP(['x'], ['y'], kernel=Kernel.squared_exponential(), mean=Mean.constant(.2, bounds={'bias': 'fixed'}), noise_variance=.05)
    held.set_training_data(X, y)
    held.setup()
    held.fit_model()
    assert held.mean.bias == .2


def test_piecewise_polynomial_self_covariance_is_degree_independent():
    """The three CasADi-only asserts of the reference's kernel tests (test_kernels.py:2720, :2954, :2977 compare the symbolic
    k(x, x) across the degrees 0..3): the covariance of a point with itself is the signal variance whatever the degree -
    checked here on numbers (isotropic, ARD, ARD with inactive dimensions)."""
    x1 = np.array([[.7]])
    x3 = np.array([[1.], [6.], [.1]])
    for spec, x in (({'type': 'piecewise_polynomial', 'kwargs': {'signal_variance': .5}}, x1),
                    ({'type': 'piecewise_polynomial', 'kwargs': {'signal_variance': .5, 'length_scales': [2., 2., 2.]}}, x3),
                    ({'type': 'piecewise_polynomial', 'kwargs': {'signal_variance': .5, 'length_scales': [2., 2.],
                                                                 'active_dims': [0, 2]}}, x3)):
        vals = []
        for degree in range(4):
            s = {'type': spec['type'], 'kwargs': dict(spec['kwargs'], degree=degree)}
            vals.append(kernel_from_spec(s)(x))
            np.testing.assert_allclose(vals[-1], ogp.kernel(s, x), rtol=1e-13)
        for v in vals[1:]:
            np.testing.assert_allclose(v, vals[0])
        np.testing.assert_allclose(vals[0], [[.5]])
