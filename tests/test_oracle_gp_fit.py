"""CPU: the oracle's hyper-parameter fit (oracle/gp_fit.py = `GaussianProcess.fit_model`, gp.py:660-697) against the
reference's own known answers (tests/test_GPs.py:846-904: optima of the GPML toolbox on the 20-point data set of
tests/test_GPs.py:835-838).  This pins, at non-trivial hyper-parameters of every kernel family, the covariance functions,
the log parametrisation (kernel.py:127-130), the log marginal likelihood (inference.py:210) and the order of
`gp.hyperparameters` = [noise variance | kernel parameters] (gp.py:408-414)."""
import numpy as np
import pytest

from oracle import gp
from oracle.gp_fit import fit

x = gp.park_miller_randn(.8, (20, 1))
y = np.sin(3 * x) + .1 * gp.park_miller_randn(.9, (20, 1))
X, Y = x.T, y.T

# (kernel, names in the reference's hyper-parameter order, fixed arguments, labels, expected, rtol)
CASES = [
    ('squared_exponential', ['length_scales', 'signal_variance'], {}, Y, [.0085251, .5298217, .8114553], 1e-5),
    ('constant', ['bias'], {}, Y + 3., [.7009480, 3.0498634], 1e-5),
    ('matern_32', ['length_scales', 'signal_variance'], {}, Y, [.0088262, .8329355, .9398366], 1e-5),
    ('matern_52', ['length_scales', 'signal_variance'], {}, Y, [.0086694, .7180206, .9571137], 1e-5),
    ('matern', ['length_scales', 'signal_variance'], {'p': 3}, Y, [.0086179, .6666981, .9421087], 1e-5),
    ('piecewise_polynomial', ['length_scales', 'signal_variance'], {'degree': 1}, Y, [.0088391, 1.6079331, .6545336], 1e-5),
    ('piecewise_polynomial', ['length_scales', 'signal_variance'], {'degree': 2}, Y, [.0088244, 2.0883167, .852502], 1e-5),
    ('piecewise_polynomial', ['length_scales', 'signal_variance'], {'degree': 3}, Y, [.0086766, 2.247363, .7873581], 1e-5),
    ('polynomial', ['signal_variance', 'offset'], {'degree': 3}, Y, [.0980796, 1.3112287, .5083423], 1e-5),
    ('linear', ['signal_variance'], {}, Y, [.6627861, .008198], 1e-4),
    ('neural_network', ['signal_variance', 'weight_variance'], {}, Y, [.0095177, 5.7756069, .1554265], 1e-5),
    ('periodic', ['signal_variance', 'length_scales', 'period'], {}, Y, [.4975112, .159969, .5905631, .8941061], 1e-3),
]


@pytest.mark.parametrize('kernel,names,fixed,labels,expected,rtol', CASES,
                         ids=[c[0] + ''.join(f'_{v}' for v in c[2].values()) for c in CASES])
def test_fit_reproduces_the_reference_optima(kernel, names, fixed, labels, expected, rtol):
    values, _ = fit(kernel, names, X, labels, noise_variance=np.exp(-2), fixed=fixed)
    np.testing.assert_allclose(values, expected, rtol=max(rtol, 2e-6))      # the expected values carry 7 digits


def test_exponential_kernel_drives_the_noise_to_zero():
    """tests/test_GPs.py:856-861: noise variance < 1e-5, the other two to 1e-5."""
    values, _ = fit('exponential', ['length_scales', 'signal_variance'], X, Y, noise_variance=np.exp(-2))
    assert values[0] < 1e-5
    np.testing.assert_allclose(values[1:], [.9769135, .5935212], rtol=1e-5)


def test_rational_quadratic_flat_direction():
    """tests/test_GPs.py:871-877: alpha runs off along a flat direction and is ignored; the others are the SE optimum."""
    values, fv = fit('rational_quadratic', ['length_scales', 'signal_variance', 'alpha'], X, Y, noise_variance=np.exp(-2))
    np.testing.assert_allclose(values[:3], [.00852509, .529822, .811455], rtol=2e-4)
    assert values[3] > 1e3
    se, fse = fit('squared_exponential', ['length_scales', 'signal_variance'], X, Y, noise_variance=np.exp(-2))
    assert abs(fv - fse) < 1e-5


# ---- the product's fit_model() driver (host logic) with the device objective replaced by the oracle's LML ----------------
_ORACLE_TYPE = {'SE': 'squared_exponential', 'Const': 'constant', 'M32': 'matern_32', 'M52': 'matern_52', 'Lin': 'linear',
                'NN': 'neural_network', 'Periodic': 'periodic', 'Poly': 'polynomial', 'PP': 'piecewise_polynomial',
                'RQ': 'rational_quadratic', 'E': 'exponential'}


def _host_fit(kernel, labels):
    """GaussianProcess.fit_model with setup()/log_marginal_likelihood() served by the oracle (no device needed)."""
    import types
    from hilo_mpc_amd import GP
    g = GP(['x'], ['y'], kernel=kernel, noise_variance=np.exp(-2))
    g.set_training_data(X, labels)

    def setup(self, device_index=None, **kw):
        k = self.kernel
        kwargs = {a: getattr(k, a) for a in k._hyper}
        if hasattr(k, 'degree'):
            kwargs['degree'] = k.degree
        self._post = gp.Posterior({'type': _ORACLE_TYPE[k.acronym], 'kwargs': kwargs}, {'type': 'zero'}, self._X_train,
                                  self._y_train, self.noise_variance)
        self._handle = object()
        self._dev = types.SimpleNamespace(index=0)
    g.setup = types.MethodType(setup, g)
    g.log_marginal_likelihood = types.MethodType(lambda self: self._post.lml, g)
    g._destroy = types.MethodType(lambda self: None, g)
    _serve_device_from_oracle(g)
    g.setup()
    g.fit_model()
    return g


def _serve_device_from_oracle(g):
    """fit_model's two device services (hilo_gp_refit, hilo_gp_lml_gradient) answered by the oracle: refit = the patched
    setup(), gradient = central differences of the oracle's LML at the free log hyper-parameters."""
    import types

    def refit(self):
        try:
            self.setup()
            return bool(np.isfinite(self._post.lml))
        except np.linalg.LinAlgError:
            return False

    def grad(self, th, h):
        out = np.zeros(th.size)
        for i in range(th.size):
            e = np.zeros_like(th)
            e[i] = h
            self._set_hyperparameters(np.exp(th + e))
            self.setup()
            up = self._post.lml
            self._set_hyperparameters(np.exp(th - e))
            self.setup()
            out[i] = (up - self._post.lml) / (2 * h)
        self._set_hyperparameters(np.exp(th))
        self.setup()
        return out
    g._device_refit = types.MethodType(refit, g)
    g._device_lml_gradient = types.MethodType(grad, g)


@pytest.mark.parametrize('make,labels,expected,rtol', [
    (lambda K: K.squared_exponential(), Y, [.0085251, .5298217, .8114553], 1e-5),
    (lambda K: K.constant(), Y + 3., [.7009480, 3.0498634], 1e-5),
    (lambda K: K.matern_52(), Y, [.0086694, .7180206, .9571137], 1e-5),
    (lambda K: K.polynomial(3), Y, [.0980796, 1.3112287, .5083423], 1e-5),
    (lambda K: K.periodic(), Y, [.4975112, .159969, .5905631, .8941061], 1e-3),
], ids=['SE', 'Const', 'M52', 'Poly', 'Periodic'])
def test_product_fit_driver_reproduces_the_reference_optima(make, labels, expected, rtol):
    import warnings
    from hilo_mpc_amd import Kernel
    with warnings.catch_warnings():
        warnings.simplefilter('error')                      # a successful fit must not warn (tests/test_GPs.py:912-916)
        g = _host_fit(make(Kernel), labels)
    np.testing.assert_allclose(g.hyperparameter_values, expected, rtol=max(rtol, 2e-6))
    assert g._optimization_stats['success'] and g.hyperparameter_names[0] == 'GP.noise_variance'


# ---- hyper-priors (tests/test_GPs.py:366-385, :596-621) ----------------------------------------------------------------
X6 = np.array([[0., .5, 1. / np.sqrt(2.), np.sqrt(3.) / 2., 1., 0.], [1., np.sqrt(3.) / 2., 1. / np.sqrt(2.), .5, 0., -1.]])
Y6 = np.array([[0., np.pi / 6., np.pi / 4., np.pi / 3., np.pi / 2., np.pi]])
SE2 = ['length_scales', 'signal_variance']


def test_lml_with_laplace_prior_kat():
    from oracle.gp_fit import negative_lml
    np.testing.assert_approx_equal(-negative_lml('squared_exponential', SE2, np.zeros(3), X6, Y6), -9.82229944)
    np.testing.assert_approx_equal(-negative_lml('squared_exponential', SE2, np.zeros(3), X6, Y6,
                                                 priors={1: ('laplace', 0., 1.)}), -10.16887303)


def test_fit_with_gaussian_prior_kat():
    y2 = Y6 + np.array([[.23757934, .55730318, .02598826, .06349002, .26647032, -.137302]])
    values, _ = fit('squared_exponential', SE2, X6, y2, noise_variance=1., priors={0: ('gaussian', .2, .01)})
    np.testing.assert_allclose(values[0], 1.406995, rtol=1e-6)
    plain, _ = fit('squared_exponential', SE2, X6, Y6, noise_variance=1.)
    assert plain[0] < 1e-3                                                  # tests/test_GPs.py:578-591


def test_product_hyperprior_driver():
    """The product's prior bookkeeping and fit driver (device objective replaced by the oracle's LML)."""
    import types
    from hilo_mpc_amd import GP
    g = GP(['x', 'y'], 'z')
    g.set_training_data(X6, Y6)

    def setup(self, device_index=None, **kw):
        k = self.kernel
        self._post = gp.Posterior({'type': 'squared_exponential', 'kwargs': {a: getattr(k, a) for a in k._hyper}},
                                  {'type': 'zero'}, self._X_train, self._y_train, self.noise_variance)
        self._handle, self._dev = object(), types.SimpleNamespace(index=0)
    g.setup = types.MethodType(setup, g)
    g._destroy = types.MethodType(lambda self: None, g)
    _serve_device_from_oracle(g)
    g.setup()

    def lml(self):
        return self._post.lml + self._log_hyperprior()
    g.log_marginal_likelihood = types.MethodType(lml, g)
    np.testing.assert_approx_equal(g.log_marginal_likelihood(), -9.82229944)
    with pytest.raises(KeyError):
        g.set_hyperprior('SE.nope', 'Laplace')
    with pytest.raises(ValueError, match="not recognized"):
        g.set_hyperprior('SE.length_scales', 'Cauchy')
    g.set_hyperprior('SE.length_scales', 'Laplace', mean=0., variance=1.)
    np.testing.assert_approx_equal(g.log_marginal_likelihood(), -10.16887303)
    g.set_hyperprior('SE.length_scales', None)
    g.set_hyperprior('GP.noise_variance', 'Gaussian', mean=.2, variance=.01)
    g.set_training_data(X6, Y6 + np.array([[.23757934, .55730318, .02598826, .06349002, .26647032, -.137302]]))
    g.setup()
    before = g.log_marginal_likelihood()
    g.fit_model()
    assert g.log_marginal_likelihood() > before
    np.testing.assert_allclose(g.noise_variance, 1.406995, rtol=1e-6)


@pytest.mark.parametrize('kernel,names,fixed', [(c[0], c[1], c[2]) for c in CASES[:1] + CASES[2:4] + CASES[8:11]],
                         ids=['SE', 'M32', 'M52', 'Poly', 'Lin', 'NN'])
def test_trace_formula_gradient_matches_finite_differences_of_the_lml(kernel, names, fixed):
    """1/2 tr((alpha alpha^T - K^-1) dK/dtheta) against central differences of the LML itself; zero at the fitted optimum."""
    from oracle.gp_fit import lml_gradient, negative_lml
    th = np.log([np.exp(-2)] + [1.3, .7, 1.1][:len(names)])
    g = lml_gradient(kernel, names, th, X, Y, fixed=fixed)
    h = 1e-6
    fd = np.array([-(negative_lml(kernel, names, th + h * e, X, Y, fixed) - negative_lml(kernel, names, th - h * e, X, Y, fixed))
                   / (2 * h) for e in np.eye(th.size)])
    np.testing.assert_allclose(g, fd, rtol=2e-6, atol=2e-7)
    values, _ = fit(kernel, names, X, Y, noise_variance=np.exp(-2), fixed=fixed)
    assert np.max(np.abs(lml_gradient(kernel, names, np.log(values), X, Y, fixed=fixed))) < 2e-4


def test_mean_hyperparameters_are_fitted_with_the_kernels():
    """gp.py:408-414: a non-fixed hyper-parameter of the mean function is an optimisation variable like the kernel's.  For a
    constant mean the optimum has a closed form given the covariance: b* = (1^T Ky^-1 y) / (1^T Ky^-1 1); `bounds={'bias':
    'fixed'}` keeps it out (mean.py:270-277)."""
    import types
    import warnings
    from hilo_mpc_amd import GP, Kernel, Mean
    assert Mean.constant(2.).trainable_hyperparameters() == ['Const.bias']
    assert Mean.constant(2., bounds={'bias': 'fixed'}).trainable_hyperparameters() == []
    assert Mean.zero().trainable_hyperparameters() == [] and Mean.one().trainable_hyperparameters() == []
    assert (Mean.constant() + Mean.polynomial(2)).trainable_hyperparameters() == ['Const.bias', 'Poly.coefficient', 'Poly.offset']
    assert Mean.linear(coefficient=[1., 2.]).trainable_hyperparameters() == ['Lin.coefficient_0', 'Lin.coefficient_1']

    def host_fit(mean):
        g = GP(['x'], ['y'], kernel=Kernel.squared_exponential(), mean=mean, noise_variance=np.exp(-2))
        g.set_training_data(X, Y + 3.)

        def setup(self, device_index=None, **kw):
            k = self.kernel
            self._post = gp.Posterior({'type': 'squared_exponential', 'kwargs': {a: getattr(k, a) for a in k._hyper}},
                                      {'type': 'constant', 'kwargs': {'bias': self.mean.bias}}, self._X_train, self._y_train,
                                      self.noise_variance)
            self._handle, self._dev = object(), types.SimpleNamespace(index=0)
        g.setup = types.MethodType(setup, g)
        g.log_marginal_likelihood = types.MethodType(lambda self: self._post.lml, g)
        g._destroy = types.MethodType(lambda self: None, g)
        _serve_device_from_oracle(g)
        g.setup()
        with warnings.catch_warnings():
            warnings.simplefilter('error')
            g.fit_model()
        return g
    g = host_fit(Mean.constant(.5))
    post = g._post
    Ky = post.R.T @ post.R
    one, yv = np.ones(Ky.shape[0]), g._y_train.ravel()
    b_star = (one @ np.linalg.solve(Ky, yv)) / (one @ np.linalg.solve(Ky, one))
    np.testing.assert_allclose(g.mean.bias, b_star, rtol=1e-5)
    assert abs(g.mean.bias - 3.) < 1.5 and g._optimization_stats['success']
    fixed = host_fit(Mean.constant(.5, bounds={'bias': 'fixed'}))
    assert fixed.mean.bias == .5 and fixed._post.lml < g._post.lml                      # the fitted bias explains the data better


