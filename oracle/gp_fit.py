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


def negative_lml(kernel_type, names, theta, X, y, fixed=None, mean_spec=None):
    """theta = logs of [noise variance | the kernel hyper-parameters `names`]."""
    kw = dict(fixed or {})
    kw.update({n: float(np.exp(t)) for n, t in zip(names, theta[1:])})
    try:
        post = Posterior({'type': kernel_type, 'kwargs': kw}, mean_spec or {'type': 'zero'}, X, y, float(np.exp(theta[0])))
    except np.linalg.LinAlgError:
        return np.inf
    return -post.lml if np.isfinite(post.lml) else np.inf


def fit(kernel_type, names, X, y, noise_variance=1., start=None, fixed=None, mean_spec=None, h=1e-6):
    """Returns (values [noise variance | names...], -LML at the optimum)."""
    start = dict(start or {})
    th0 = np.log([noise_variance] + [start.get(n, 1.) for n in names])

    def f(th):
        return negative_lml(kernel_type, names, th, X, y, fixed, mean_spec)

    def g(th):
        out = np.zeros_like(th)
        for i in range(th.size):
            e = np.zeros_like(th)
            e[i] = h
            out[i] = (f(th + e) - f(th - e)) / (2 * h)
        return out
    res = minimize(f, th0, jac=g, method='BFGS', options={'gtol': 1e-8, 'maxiter': 1000})
    return np.exp(res.x), float(res.fun)
