"""Pin oracle/gp.py against the reference's known-answer tests (tests/test_kernels.py, test_means.py, test_GPs.py)."""
import json
import os

import numpy as np
import pytest

from oracle import gp

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
KCASES = json.load(open(os.path.join(GOLD, 'kernels_kat.json')))
MCASES = json.load(open(os.path.join(GOLD, 'means_kat.json')))


@pytest.mark.parametrize('case', KCASES, ids=[f"{c['ref_test'].split('::')[-1]}@{c['ref_line']}" for c in KCASES])
def test_kernel_kat(case):
    K = gp.kernel(case['spec'], *[np.array(a) for a in case['args']])
    np.testing.assert_allclose(K, np.array(case['expected']), rtol=case['tol'].get('rtol', 1e-7),
                               atol=case['tol'].get('atol', 0))


@pytest.mark.parametrize('case', MCASES, ids=[f"{c['ref_test'].split('::')[-1]}@{c['ref_line']}" for c in MCASES])
def test_mean_kat(case):
    mu = gp.mean(case['spec'], *[np.array(a) for a in case['args']])
    np.testing.assert_allclose(mu, np.array(case['expected']), rtol=case['tol'].get('rtol', 1e-7),
                               atol=case['tol'].get('atol', 0))


def test_gp_lml_kat():
    # reference tests/test_GPs.py:326-329, 354-363: defaults SE l=1, s_f^2=1, s_n^2=1, zero mean
    X = np.array([[0., .5, 1. / np.sqrt(2.), np.sqrt(3.) / 2., 1., 0.],
                  [1., np.sqrt(3.) / 2., 1. / np.sqrt(2.), .5, 0., -1.]])
    y = np.array([0., np.pi / 6., np.pi / 4., np.pi / 3., np.pi / 2., np.pi])
    post = gp.Posterior({'type': 'squared_exponential'}, {'type': 'zero'}, X, y, 1.)
    np.testing.assert_approx_equal(post.lml, -9.82229944)


def test_gp_rasmussen_lml_kat():
    # reference tests/test_GPs.py:1102-1123 (Matern-3/2 l=.25, mean .5 x + 1, s_n^2 = .01)
    ker = {'type': 'matern_32', 'kwargs': {'length_scales': .25}}
    mu = {'type': 'sum', 'children': [{'type': 'linear', 'kwargs': {'coefficient': .5}}, {'type': 'one'}]}
    x = gp.park_miller_randn(.3, (20, 1))
    K = gp.kernel(ker, x.T)
    m = gp.mean(mu, x.T)
    y = np.linalg.cholesky(K) @ gp.park_miller_randn(.15, (20, 1)) + m.T + .1 * gp.park_miller_randn(.2, (20, 1))
    post = gp.Posterior(ker, mu, x.T, y.T, .1 ** 2)
    np.testing.assert_approx_equal(post.lml, -11.9706317)
    # reference tests/test_GPs.py:654-666 properties: noise-free mean equal, variance smaller
    xs = np.linspace(-1.9, 1.9, 101).reshape(1, -1)
    m1, v1 = post.predict(xs)
    m2, v2 = post.predict(xs, noise_free=True)
    np.testing.assert_allclose(m1, m2)
    assert np.all(v2 < v1) and np.all(v2 > -1e-12)


def test_piecewise_polynomial_self_covariance_is_degree_independent():
    """test_kernels.py:2720, :2954, :2977 (CasADi-only asserts: the symbolic k(x, x) is the same for the degrees 0..3)."""
    x3 = np.array([[1.], [6.], [.1]])
    for kw, x in (({'signal_variance': .5}, np.array([[.7]])), ({'signal_variance': .5, 'length_scales': [2., 2., 2.]}, x3),
                  ({'signal_variance': .5, 'length_scales': [2., 2.], 'active_dims': [0, 2]}, x3)):
        vals = [gp.kernel({'type': 'piecewise_polynomial', 'kwargs': dict(kw, degree=d)}, x) for d in range(4)]
        for v in vals:
            np.testing.assert_allclose(v, [[.5]])
