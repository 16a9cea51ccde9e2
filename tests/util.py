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


def golden_or_compute(name, compute):
    """Oracle results that cost minutes of symbolic algebra: `compute()` returns a (nested) dict of arrays, numbers and index lists;
    it is kept as tests/golden/<name>.json by tests/golden/make_mhe_gen_fixtures.py (which calls the same `compute`) and loaded from
    there when the fixture exists - the test then compares the device result with the SAME oracle numbers without re-deriving
    them on the GPU box's CPU.  Arrays are stored as nested lists of repr-exact doubles."""
    import json
    import os
    import numpy as np
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name + '.json')

    def dec(v):
        if isinstance(v, dict):
            if set(v) == {'__array__', 'dtype'}:
                return np.asarray(v['__array__'], dtype=v['dtype'])
            return {k: dec(x) for k, x in v.items()}
        return v
    if os.path.exists(path) and not os.environ.get('HILO_RECOMPUTE_GOLDEN'):
        return dec(json.load(open(path)))
    return compute()


def golden_dump(name, data):
    import json
    import os
    import numpy as np

    def enc(v):
        if isinstance(v, dict):
            return {k: enc(x) for k, x in v.items()}
        if isinstance(v, np.ndarray):
            return {'__array__': v.tolist(), 'dtype': str(v.dtype)}
        if isinstance(v, (np.integer,)):
            return int(v)
        if isinstance(v, (np.floating,)):
            return float(v)
        if isinstance(v, (list, tuple)):
            return [enc(x) for x in v]
        return v
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name + '.json')
    json.dump(enc(data), open(path, 'w'))
    return path
