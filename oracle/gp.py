"""Oracle: covariance functions, mean functions and exact GP inference, plain numpy.

TEST INFRASTRUCTURE ONLY - never imported by the product package.

Follows the reference (HILO-MPC v1.1.0, `hilo_mpc/modules/machine_learning/gp/`):

* kernel.py:97-140    `Kernel.__call__`: X is (n_features x n_obs); hyper-parameters enter as logs, and
                      anything named `*variance*` as log(value)/2, i.e. a log standard deviation (:127-130)
* kernel.py:179-205   covariance matrix K[i,j] = k(X[:,i], Xbar[:,j])
* kernel.py:538-555   length-scale matrix M = diag(exp(-2 log l)) over the active dimensions
* kernel.py:465-485   Constant            exp(2 log b)
* kernel.py:650-701   gamma-exponential   exp(2 log s - alpha d2^(p/2)); SE: p = 2, alpha = 1/2 (:733-734)
* kernel.py:783-826   Matern nu=p+1/2     exp(2 log s - d) f(d), d = sqrt(2 nu d2), f by the Horner recursion there
* kernel.py:972-1003  rational quadratic  exp(2 log s)(1 + d2/(2 alpha))^(-alpha)
* kernel.py:1069-1109 piecewise polynomial (compact support), j = floor(D/2) + q + 1
* kernel.py:1202-1234 polynomial          exp(2 log s)(x.xbar + offset)^p ; linear: p = 1, offset = 0 (:1258-1259)
* kernel.py:1309-1332 neural network      exp(2 log s) asin((1 + x.xbar)/sqrt((w+1+x.x)(w+1+xbar.xbar)))
* kernel.py:1394-1423 periodic            exp(2 log s - 2 (sin(pi (x-xbar)/T)/l)^2)   (1 active dimension)
* kernel.py:1562-1666 Sum, Product, Power (Power evaluates its child at (x, x) - `self.kernel_1(x)`, :1651)
* mean.py:280-305, 422-470, 624-766   Constant/Zero/One, Polynomial/Linear (M^T x + offset)^p, Scale/Sum/Product/Power
* inference.py:172-221 `ExactInference.get_posterior`: no jitter; L = chol(s_n^2 I + K) (upper, R^T R);
                      alpha = R \\ (R^T \\ (y-m)); LML = -1/2 (y-m) alpha - sum log diag R - n/2 log 2 pi;
                      mu* = m(x*) + k*^T alpha; v = R^T \\ k*; var* = k** - v^T v
* gp.py:699-718       `predict`: var += s_n^2 unless noise_free

Kernels and means are described by plain spec dicts (the JSON form produced by
`tests/golden/make_kernel_golden.py`): {'type': ..., 'kwargs': {...}, 'children': [...]}.
"""
from __future__ import annotations

from math import factorial, gamma as gamma_fun

import numpy as np
from scipy.linalg import solve_triangular


def _kw(spec):
    return dict(spec.get('kwargs', {}))


def _active(kw, D):
    ad = kw.get('active_dims')
    if ad is None:
        return np.arange(D)
    return np.atleast_1d(np.asarray(ad, dtype=int))


def _log_param(value, is_variance):
    """kernel.py:127-130: parameter.log / 2 if 'variance' in the name else parameter.log."""
    with np.errstate(divide='ignore'):
        lg = np.log(np.asarray(value, dtype=float))
    return lg / 2. if is_variance else lg


def _M(kw, n_active):
    """kernel.py:538-555 (isotropic if the length scale is a scalar)."""
    ls = kw.get('length_scales', 1.)
    ard = kw.get('ard', False)
    if np.ndim(ls) == 0 and ard:
        ls = n_active * [ls]
    log_l = _log_param(ls, False)
    if np.ndim(log_l) == 0:
        return np.exp(-2 * log_l) * np.ones(n_active)
    if len(log_l) != n_active:
        raise ValueError("Length scales vector dimension does not equal input space dimension.")
    return np.exp(-2 * log_l)


def _d2(X, Xb, ad, Mdiag):
    diff = X[ad][:, :, None] - Xb[ad][:, None, :]
    return np.einsum('d,dij->ij', Mdiag, diff * diff)


def matern_poly(p):
    """Coefficients of the Horner recursion of kernel.py:800-815."""
    if p > 1:
        g1, g2 = gamma_fun(p + 1), gamma_fun(2 * p + 1)
        poly = [g1 / g2 * factorial(p + k) / (factorial(k) * factorial(p - k)) * 2. ** (p - k) for k in range(p - 1)]
    else:
        poly = []
    if p == 0:
        poly.append(0.)
    elif p >= 1:
        poly.append(1.)
    for k in range(len(poly) - 1, 0, -1):
        poly[k - 1] /= poly[k]
    return poly


def kernel(spec, X, Xbar=None):
    """Covariance matrix K (n_obs(X) x n_obs(Xbar)) for feature-major inputs (kernel.py:97-140)."""
    X = np.atleast_2d(np.asarray(X, dtype=float))
    Xb = X if Xbar is None else np.atleast_2d(np.asarray(Xbar, dtype=float))
    assert X.shape[0] == Xb.shape[0], "X and X_bar do not have the same input space dimensions"
    D = X.shape[0]
    t = spec['type']
    kw = _kw(spec)
    ch = spec.get('children', [])
    n, m = X.shape[1], Xb.shape[1]

    if t == 'sum':
        return kernel(ch[0], X, Xb) + kernel(ch[1], X, Xb)
    if t == 'product':
        return kernel(ch[0], X, Xb) * kernel(ch[1], X, Xb)
    if t == 'power':
        # kernel.py:1651: K = self.kernel_1(x) -> k(x, x), independent of x_bar
        kxx = np.array([kernel(ch[0], X[:, [i]], X[:, [i]])[0, 0] for i in range(n)])
        return np.repeat((kxx ** kw['power'])[:, None], m, axis=1)

    ad = _active(kw, D)
    if t == 'constant':
        lb = _log_param(kw.get('bias', 1.), False)
        return np.full((n, m), np.exp(2 * lb))

    ls2 = 2 * _log_param(kw.get('signal_variance', 1.), True)     # 2 log sigma
    if t in ('squared_exponential', 'gamma_exponential'):
        if t == 'squared_exponential':
            p, alpha = 2., .5
        else:
            g = kw.get('gamma', 1.)
            g = g / (2 - g)
            p = 2 / (1 - np.exp(-np.log(g)))
            alpha = kw.get('alpha', 1.) if kw.get('alpha') is not None else 1.
        d2 = _d2(X, Xb, ad, _M(kw, ad.size))
        return np.exp(ls2 - alpha * d2 ** (p / 2))
    if t in ('exponential', 'matern_32', 'matern_52', 'matern'):
        p = {'exponential': 0, 'matern_32': 1, 'matern_52': 2}.get(t, kw.get('p'))
        nu = p + .5
        d = np.sqrt(2 * nu * _d2(X, Xb, ad, _M(kw, ad.size)))
        poly = matern_poly(p)
        f = 1. + d * poly[0]
        for k in range(1, len(poly)):
            f = 1. + d * poly[k] * f
        return np.exp(ls2 - d) * f
    if t == 'rational_quadratic':
        la = _log_param(kw.get('alpha', 1.), False)
        d2 = _d2(X, Xb, ad, _M(kw, ad.size))
        return np.exp(ls2) * (1 + .5 * d2 / np.exp(la)) ** (-np.exp(la))
    if t == 'piecewise_polynomial':
        q = kw['degree']
        j = np.floor(ad.size / 2) + q + 1
        d2 = _d2(X, Xb, ad, _M(kw, ad.size))
        d = np.sqrt(d2)
        if q == 0:
            f = 1.
        elif q == 1:
            f = (j + 1) * d + 1
        elif q == 2:
            f = (j ** 2 + 4 * j + 3) / 3 * d2 + (j + 2) * d + 1
        elif q == 3:
            f = (j ** 3 + 9 * j ** 2 + 23 * j + 15) / 15 * d ** 3 + (6 * j ** 2 + 36 * j + 45) / 15 * d2 + (j + 3) * d + 1
        else:
            raise RuntimeError("The parameter 'q' has to be one of the following integers: 0, 1, 2, 3")
        return np.exp(ls2) * (d < 1.) * np.fmax(1. - d, 0.) ** (j + q) * f
    if t in ('polynomial', 'linear'):
        if t == 'linear':
            p, off = 1, 0.
        else:
            p, off = kw['degree'], kw.get('offset', 1.)
        lo = _log_param(off, False)
        return np.exp(ls2) * (X[ad].T @ Xb[ad] + np.exp(lo)) ** p
    if t == 'neural_network':
        lw2 = 2 * _log_param(kw.get('weight_variance', 1.), True)
        num = 1. + X[ad].T @ Xb[ad]
        den1 = np.sqrt(np.exp(lw2) + 1. + np.sum(X[ad] * X[ad], axis=0))
        den2 = np.sqrt(np.exp(lw2) + 1. + np.sum(Xb[ad] * Xb[ad], axis=0))
        return np.exp(ls2) * np.arcsin(num / (den1[:, None] * den2[None, :]))
    if t == 'periodic':
        if ad.size != 1:
            raise ValueError("periodic kernel: the reference expression is scalar only for one active dimension")
        ll = _log_param(kw.get('length_scales', 1.), False)
        lp = _log_param(kw.get('period', 1.), False)
        arg = np.sin(np.pi * (X[ad[0]][:, None] - Xb[ad[0]][None, :]) / np.exp(lp)) / np.exp(ll)
        return np.exp(ls2 - 2 * arg ** 2)
    raise ValueError(f"unknown kernel type {t}")


def mean(spec, X):
    """Mean function values (1 x n_obs) for feature-major X (mean.py:90-116)."""
    X = np.atleast_2d(np.asarray(X, dtype=float))
    D, n = X.shape
    t = spec['type']
    kw = _kw(spec)
    ch = spec.get('children', [])
    if t == 'sum':
        return mean(ch[0], X) + mean(ch[1], X)
    if t == 'product':
        return mean(ch[0], X) * mean(ch[1], X)
    if t == 'power':
        return mean(ch[0], X) ** kw['power']
    if t == 'scale':
        return kw['scale'] * mean(ch[0], X)
    if t in ('constant', 'zero', 'one'):
        b = {'zero': 0., 'one': 1.}.get(t, kw.get('bias', 1.))
        return np.full((1, n), float(b))
    if t in ('polynomial', 'linear'):
        ad = _active(kw, D)
        if t == 'linear':
            p, off = 1, 0.
        else:
            p, off = kw['degree'], kw.get('offset', 1.)
        c = np.asarray(kw.get('coefficient', 1.), dtype=float)
        if c.ndim == 0:
            c = c * np.ones(ad.size)
        if c.size != ad.size:
            raise ValueError("Coefficient vector dimension does not equal input space dimension.")
        return ((c @ X[ad] + off) ** p)[None, :]
    raise ValueError(f"unknown mean type {t}")


class Posterior:
    """Everything `ExactInference.get_posterior` produces for one training set (inference.py:197-217)."""

    def __init__(self, kernel_spec, mean_spec, X, y, noise_variance):
        X = np.atleast_2d(np.asarray(X, dtype=float))
        y = np.asarray(y, dtype=float).reshape(-1)
        n = X.shape[1]
        log_sn = _log_param(noise_variance, True)
        sn2 = np.exp(2 * log_sn)
        K = kernel(kernel_spec, X, X)
        self.R = np.linalg.cholesky(sn2 * np.eye(n) + K).T            # upper, R^T R = K + sn2 I
        self.ym = y - mean(mean_spec, X)[0]
        self.alpha = solve_triangular(self.R, solve_triangular(self.R.T, self.ym, lower=True), lower=False)
        self.lml = -.5 * self.ym @ self.alpha - np.sum(np.log(np.diag(self.R))) - n / 2 * np.log(2 * np.pi)
        self.X, self.sn2, self.kernel_spec, self.mean_spec = X, sn2, kernel_spec, mean_spec

    def predict(self, Xq, noise_free=False):
        """gp.py:699-718: returns (mean (1 x m), var (1 x m)); every query column is treated on its own."""
        Xq = np.atleast_2d(np.asarray(Xq, dtype=float))
        Ks = kernel(self.kernel_spec, self.X, Xq)
        mu = mean(self.mean_spec, Xq)[0] + Ks.T @ self.alpha
        v = solve_triangular(self.R.T, Ks, lower=True)
        kss = np.array([kernel(self.kernel_spec, Xq[:, [i]], Xq[:, [i]])[0, 0] for i in range(Xq.shape[1])])
        var = kss - np.sum(v * v, axis=0)
        if not noise_free:
            var = var + self.sn2
        return mu[None, :], var[None, :]


def park_miller_randn(seed, shape):
    """Deterministic normal deviates: Park-Miller minimal-standard LCG (a=7^5, m=2^31-1, Schrage's
    factorisation) feeding a Box-Muller transform - the generator the GPML demos use, restated here to rebuild
    the data set of the reference's Rasmussen regression test (tests/test_GPs.py:1102-1110)."""
    n = int(np.prod(shape))
    N = int(np.ceil(n / 2) * 2)
    a, m = 7 ** 5, 2 ** 31 - 1
    q, r = m // a, m % a
    s = int(np.fix(seed * 2 ** 31))
    u = np.empty(N)
    for k in range(N):
        s = a * (s % q) - r * (s // q)
        if s < 0:
            s += m
        u[k] = s / 2 ** 31
    h = N // 2
    w = np.sqrt(-2 * np.log(u[:h]))
    x = np.concatenate([w * np.cos(2 * np.pi * u[h:]), w * np.sin(2 * np.pi * u[h:])])
    return x[:n].reshape(shape)
