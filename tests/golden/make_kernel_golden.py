#!/usr/bin/env python3
"""Extract the known-answer vectors of the reference's kernel / mean / GP tests into JSON fixtures.

Run in the build container only (needs /root/reference):

    python tests/golden/make_kernel_golden.py

It does NOT import the reference package (casadi is not installable here) and copies no reference
source: it walks the AST of `/root/reference/tests/test_kernels.py` and `test_means.py`, and for each
test function that ends in `np.testing.assert_allclose(result, expected)` records *data only* -
the constructor name + keyword arguments of the object under test, the numeric inputs and the expected
array - into `tests/golden/kernels_kat.json` / `means_kat.json`.
"""
import ast
import json
import os
import sys

import numpy as np

REF_TESTS = '/root/reference/tests'
HERE = os.path.dirname(os.path.abspath(__file__))

KERNEL_CLASSES = {
    'ConstantKernel': 'constant', 'SquaredExponentialKernel': 'squared_exponential',
    'ExponentialKernel': 'exponential', 'Matern32Kernel': 'matern_32', 'Matern52Kernel': 'matern_52',
    'RationalQuadraticKernel': 'rational_quadratic', 'PiecewisePolynomialKernel': 'piecewise_polynomial',
    'PolynomialKernel': 'polynomial', 'LinearKernel': 'linear', 'NeuralNetworkKernel': 'neural_network',
    'PeriodicKernel': 'periodic',
}
MEAN_CLASSES = {
    'ConstantMean': 'constant', 'ZeroMean': 'zero', 'OneMean': 'one', 'PolynomialMean': 'polynomial',
    'LinearMean': 'linear',
}
POSITIONAL = {'piecewise_polynomial': ['degree'], 'polynomial': ['degree'], 'constant': ['bias']}
MEAN_POSITIONAL = {'polynomial': ['degree'], 'constant': ['bias']}


def _clean(v):
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, (list, tuple)):
        return [_clean(i) for i in v]
    return v


class Spec:
    """Records how an object was built; supports the operator overloads the tests use."""

    def __init__(self, type_, kwargs=None, children=None):
        self.type = type_
        self.kwargs = kwargs or {}
        self.children = children or []

    def to_json(self):
        d = {'type': self.type}
        if self.kwargs:
            d['kwargs'] = {k: _clean(v) for k, v in self.kwargs.items()}
        if self.children:
            d['children'] = [c.to_json() for c in self.children]
        return d

    def __add__(self, o):
        return Spec('sum', children=[self, o]) if isinstance(o, Spec) else NotImplemented

    def __mul__(self, o):
        if isinstance(o, Spec):
            return Spec('product', children=[self, o])
        return Spec('scale', {'scale': o}, [self])

    __rmul__ = __mul__

    def __pow__(self, p):
        return Spec('power', {'power': p}, [self])

    def __call__(self, *args):
        return Call(self, args)

    def __getattr__(self, item):          # attribute reads in the tests (e.g. kernel.length_scales) -> inert
        raise AttributeError(item)


class Call:
    def __init__(self, spec, args):
        self.spec = spec
        self.args = args


def factory(table, default_positional):
    class F:
        pass

    def mk(name):
        def ctor(*args, **kwargs):
            pos = POSITIONAL.get(name, []) if default_positional else []
            for a, n in zip(args, pos):
                kwargs[n] = a
            if len(args) > len(pos):
                raise ValueError('unexpected positional')
            return Spec(name, kwargs)
        return ctor

    ns = {}
    for cls, name in table.items():
        ns[cls] = mk(name)
        setattr(F, name, staticmethod(mk(name)))
    return F, ns


def extract(path, table, facade_name):
    src = open(path).read()
    tree = ast.parse(src)
    Facade, class_ns = factory(table, True)
    cases = []
    for cls in [c for c in tree.body if isinstance(c, ast.ClassDef)]:
        for fn in [f for f in cls.body if isinstance(f, ast.FunctionDef)]:
            seg = ast.get_source_segment(src, fn)
            if 'assert_allclose' not in seg or 'ca.' in seg:
                continue
            env = {'np': np, facade_name: Facade, **class_ns}

            def visit(body):
                for st in body:
                    if isinstance(st, (ast.Import, ast.ImportFrom)):
                        continue
                    if isinstance(st, ast.For):
                        # known answers asserted inside a loop (`for k in range(4): ... assert_allclose(kernel(x), table[k])`):
                        # one case per pass, with the loop variable bound
                        try:
                            values = list(eval(compile(ast.Expression(st.iter), '<ref-test>', 'eval'), env))
                        except Exception:
                            continue
                        for v in values:
                            env[st.target.id] = v
                            visit(st.body)
                        continue
                    is_assert = (isinstance(st, ast.Expr) and isinstance(st.value, ast.Call)
                                 and ast.unparse(st.value.func) == 'np.testing.assert_allclose')
                    if is_assert:
                        try:
                            got = eval(compile(ast.Expression(st.value.args[0]), '<ref-test>', 'eval'), env)
                            exp = eval(compile(ast.Expression(st.value.args[1]), '<ref-test>', 'eval'), env)
                            if isinstance(exp, Call):
                                continue
                            kw = {k.arg: eval(compile(ast.Expression(k.value), '<ref-test>', 'eval'), env)
                                  for k in st.value.keywords}
                        except Exception as e:                       # noqa
                            print(f'  skip assert in {cls.name}.{fn.name}: {e}', file=sys.stderr)
                            continue
                        if not isinstance(got, Call) or isinstance(exp, Call):
                            continue
                        cases.append({
                            'ref_test': f'{os.path.basename(path)}::{cls.name}::{fn.name}',
                            'ref_line': st.lineno,
                            'spec': got.spec.to_json(),
                            'args': [_clean(np.asarray(a, dtype=float)) for a in got.args],
                            'expected': _clean(np.asarray(exp, dtype=float)),
                            'tol': {k: float(v) for k, v in kw.items()},
                        })
                        continue
                    if isinstance(st, ast.Assign):
                        try:
                            exec(compile(ast.Module([st], []), '<ref-test>', 'exec'), env)
                        except Exception:
                            pass

            visit(fn.body)
    return cases


def main():
    k = extract(os.path.join(REF_TESTS, 'test_kernels.py'), KERNEL_CLASSES, 'Kernel')
    with open(os.path.join(HERE, 'kernels_kat.json'), 'w') as f:
        json.dump(k, f, indent=0)
    print(f'{len(k)} kernel known-answer cases')
    m = extract(os.path.join(REF_TESTS, 'test_means.py'), MEAN_CLASSES, 'Mean')
    with open(os.path.join(HERE, 'means_kat.json'), 'w') as f:
        json.dump(m, f, indent=0)
    print(f'{len(m)} mean known-answer cases')


if __name__ == '__main__':
    main()
