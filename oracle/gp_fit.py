"""Oracle: hyper-parameter fit of the exact GP, plain numpy / scipy.

TEST INFRASTRUCTURE ONLY - never imported by the product package.

`GaussianProcess.fit_model` of the reference (gp.py:660-697) minimises the negative log marginal likelihood
(inference.py:210) over the logarithms of the free hyper-parameters [noise variance | mean | kernel] (gp.py:408-414;
kernel.py:127-130: the optimisation variables are the logs).  Restated here with a quasi-Newton method on central
differences of `oracle.gp.Posterior.lml`; the optimum does not depend on the solver, which is what the reference's own
known answers (tests/test_GPs.py:846-904, taken from the GPML toolbox) pin.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import minimize

from oracle.gp import Posterior


def log_prior(kind, w, mean, variance, nu=None):
    """Log density of a hyper-prior at the optimisation variable w (util/probability.py:61-118; gp.py:553-559: the prior is
    evaluated at log(value), or log(value)/2 for a `*variance*` parameter, and ADDED to the log marginal likelihood)."""
    from math import lgamma
    kind = kind.lower()
    if kind == 'gaussian':
        return -(w - mean) ** 2 / (2. * variance) - np.log(2. * np.pi * variance) / 2.
    if kind == 'laplace':
        b = np.sqrt(variance / 2.)
        return -abs(w - mean) / b - np.log(2. * b)
    if kind in ('students_t', "student's_t", 'studentst'):
        log_z = lgamma((nu + 1.) / 2.) - lgamma(nu / 2.) - np.log(variance * (nu - 2.) * np.pi) / 2.
        return log_z - (nu + 1.) / 2. * np.log(1. + (w - mean) ** 2 / (variance * (nu - 2.)))
    raise ValueError(f"unknown prior {kind}")


def negative_lml(kernel_type, names, theta, X, y, fixed=None, mean_spec=None, priors=None):
    """theta = logs of [noise variance | the kernel hyper-parameters `names`]."""
    kw = dict(fixed or {})
    kw.update({n: float(np.exp(t)) for n, t in zip(names, theta[1:])})
    try:
        post = Posterior({'type': kernel_type, 'kwargs': kw}, mean_spec or {'type': 'zero'}, X, y, float(np.exp(theta[0])))
    except np.linalg.LinAlgError:
        return np.inf
    if not np.isfinite(post.lml):
        return np.inf
    # theta holds log(value); the reference's variable of a `*variance*` parameter is log(value) / 2
    all_names = ['noise_variance'] + list(names)
    lp = sum(log_prior(kind, theta[i] / 2. if 'variance' in all_names[i] else theta[i], *args)
             for i, (kind, *args) in (priors or {}).items())
    return -(post.lml + lp)


def fit(kernel_type, names, X, y, noise_variance=1., start=None, fixed=None, mean_spec=None, h=1e-6, priors=None):
    """Returns (values [noise variance | names...], -(LML + log priors) at the optimum).  priors: {position in
    [noise variance | names]: (kind, mean, variance[, nu])}."""
    start = dict(start or {})
    th0 = np.log([noise_variance] + [start.get(n, 1.) for n in names])

    def f(th):
        return negative_lml(kernel_type, names, th, X, y, fixed, mean_spec, priors)

    def g(th):
        out = np.zeros_like(th)
        for i in range(th.size):
            e = np.zeros_like(th)
            e[i] = h
            out[i] = (f(th + e) - f(th - e)) / (2 * h)
        return out
    res = minimize(f, th0, jac=g, method='BFGS', options={'gtol': 1e-8, 'maxiter': 1000})
    return np.exp(res.x), float(res.fun)


def lml_gradient(kernel_type, names, theta, X, y, fixed=None, mean_spec=None, h=1e-6):
    """d LML / d theta by the trace formula (Rasmussen & Williams eq. 5.9): 1/2 tr((alpha alpha^T - K_y^-1) dK_y/dtheta_i),
    K_y = K + sn2 I, with dK/dtheta_i of the KERNEL MATRIX by central differences (the covariance functions are cheap;
    the factorisation is reused for every i).  This is the shape the device gradient kernel will have: one factorisation,
    one kernel-matrix derivative per hyper-parameter, one trace."""
    from oracle.gp import kernel
    from scipy.linalg import cho_solve

    def K_of(th):
        kw = dict(fixed or {})
        kw.update({n: float(np.exp(t)) for n, t in zip(names, th[1:])})
        Xa = np.atleast_2d(np.asarray(X, dtype=float))
        return kernel({'type': kernel_type, 'kwargs': kw}, Xa, Xa) + np.exp(theta_noise(th)) * np.eye(Xa.shape[1])

    def theta_noise(th):
        return th[0]                                      # theta_0 = log(noise variance)
    theta = np.asarray(theta, dtype=float)
    kw = dict(fixed or {})
    kw.update({n: float(np.exp(t)) for n, t in zip(names, theta[1:])})
    post = Posterior({'type': kernel_type, 'kwargs': kw}, mean_spec or {'type': 'zero'}, X, y, float(np.exp(theta[0])))
    n = post.R.shape[0]
    Kinv = cho_solve((post.R, False), np.eye(n))
    A = np.outer(post.alpha, post.alpha) - Kinv
    g = np.zeros_like(theta)
    for i in range(theta.size):
        e = np.zeros_like(theta)
        e[i] = h
        dK = (K_of(theta + e) - K_of(theta - e)) / (2 * h)
        g[i] = 0.5 * np.sum(A * dK)                       # tr(A dK), both symmetric
    return g
