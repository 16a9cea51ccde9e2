"""Test helpers: build product objects from the JSON spec dicts used by the golden fixtures / oracle."""
import numpy as np


def kernel_from_spec(spec):
    from hilo_mpc_amd import Kernel
    t = spec['type']
    kw = dict(spec.get('kwargs', {}))
    ch = spec.get('children', [])
    if t == 'sum':
        return kernel_from_spec(ch[0]) + kernel_from_spec(ch[1])
    if t == 'product':
        return kernel_from_spec(ch[0]) * kernel_from_spec(ch[1])
    if t == 'power':
        return kernel_from_spec(ch[0]) ** kw['power']
    if t in ('piecewise_polynomial', 'polynomial'):
        deg = kw.pop('degree')
        return getattr(Kernel, t)(deg, **kw)
    return getattr(Kernel, t)(**kw)


def mean_from_spec(spec):
    from hilo_mpc_amd import Mean
    t = spec['type']
    kw = dict(spec.get('kwargs', {}))
    ch = spec.get('children', [])
    if t == 'sum':
        return mean_from_spec(ch[0]) + mean_from_spec(ch[1])
    if t == 'product':
        return mean_from_spec(ch[0]) * mean_from_spec(ch[1])
    if t == 'power':
        return mean_from_spec(ch[0]) ** kw['power']
    if t == 'scale':
        return kw['scale'] * mean_from_spec(ch[0])
    if t == 'polynomial':
        deg = kw.pop('degree')
        return Mean.polynomial(deg, **kw)
    return getattr(Mean, t)(**kw)


def spd_batch(rng, B, n, scale=.1, diag=.5):
    A = rng.normal(size=(B, n, n)) * scale
    return A @ np.swapaxes(A, 1, 2) + np.eye(n) * diag
