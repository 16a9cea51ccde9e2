"""Symbolic model derivatives of the engine's derivative phase (hilo_mpc_amd/symdiff.py -> csrc/hilo_models_sym.h) against the
oracle's sympy derivatives of the same right-hand sides (oracle/shooting.py), and the committed header against its generator."""
import os

import numpy as np
import pytest

from hilo_mpc_amd import zoo_expr
from hilo_mpc_amd.model import Model
from hilo_mpc_amd.symdiff import derivative_dag
from oracle import models
from oracle.shooting import ShootingMap

POINTS = {'chemostat4': ([.5, 20., 1., .5], [.3, .2], [40., 4., 1., .3]),
          'pendulum4': ([.4, -.3, .7, 1.2], [2.], []),
          'cstr3': ([.5, .5, 430.], [5e4], [])}


@pytest.mark.parametrize('name', sorted(POINTS))
def test_symbolic_jacobian_and_contracted_hessian_vs_sympy(name):
    m = zoo_expr.define(Model(name=name + '_expr'), name)
    g, f, J, H, _ = derivative_dag(m.n_x, m.n_u, m._ode)
    sm = ShootingMap(models.get(name), 1)
    rng = np.random.default_rng(3)
    nz = m.n_x + m.n_u
    for _ in range(5):
        x0, u0, p0 = POINTS[name]
        x = np.array(x0) * (1 + .2 * rng.uniform(-1, 1, len(x0)))
        u = np.array(u0) * (1 + .2 * rng.uniform(-1, 1, len(u0)))
        p = np.array(p0)
        kb = rng.normal(size=m.n_x)
        fr, Jr, Hr = (a[0] for a in sm._rhs(x[None], u[None], p[None] if len(p0) else np.zeros((1, 0)), np.array([[1.]])))
        out = g.evaluate(f + [J[a][b] for a in range(m.n_x) for b in range(nz)] + H, x, u, p, kb)
        np.testing.assert_allclose(out[:m.n_x], fr, rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(np.array(out[m.n_x:m.n_x + m.n_x * nz]).reshape(m.n_x, nz), Jr, rtol=1e-12, atol=1e-13)
        Hc = np.einsum('m,mab->ab', kb, Hr)
        Hs = np.zeros((nz, nz))
        q = m.n_x + m.n_x * nz
        for i in range(nz):
            for j in range(i + 1):
                Hs[i, j] = Hs[j, i] = out[q]
                q += 1
        np.testing.assert_allclose(Hs, Hc, rtol=1e-11, atol=1e-12 * max(1., np.abs(Hc).max()))


def test_committed_header_is_what_the_generator_writes():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    import gen_model_sym
    assert open(os.path.join(root, 'hilo_mpc_amd', 'csrc', 'hilo_models_sym.h')).read() == gen_model_sym.generate(), \
        "run tools/gen_model_sym.py"


def test_composed_functions_and_general_powers():
    """tan / sinh / cosh / tanh and a^b are composed of the device's operations (hilo_mpc_amd/expr.py): values, the symbolic
    derivative DAG and the expression-level derivative against sympy."""
    import math
    import numpy as np
    import sympy as sp
    from hilo_mpc_amd import Model, expr
    from hilo_mpc_amd.symdiff import derivative_dag
    from tests.problems import eval_exprs
    m = Model(name='fun')
    x, u = m.set_dynamical_states(['a', 'b']), m.set_inputs(['c'])
    m.set_dynamical_equations([expr.tan(x[0]) + expr.tanh(x[1] * u[0]) + x[0] ** 2.5,
                               expr.sinh(x[0]) * expr.cosh(x[1]) + (1. + x[1] * x[1]) ** x[0] + 2. ** u[0]])
    m2 = Model(name='fun2')
    m2.set_dynamical_states(['a', 'b']), m2.set_inputs(['c'])
    m2.set_dynamical_equations(['tan(a) + tanh(b * c) + a ^ 2.5', 'sinh(a) * cosh(b) + (1. + b * b) ** a + 2. ** c'])
    a, b, c = sp.symbols('a b c')
    f = sp.Matrix([sp.tan(a) + sp.tanh(b * c) + a ** 2.5, sp.sinh(a) * sp.cosh(b) + (1 + b * b) ** a + 2 ** c])
    J = sp.lambdify([a, b, c], f.jacobian([a, b, c]))
    F = sp.lambdify([a, b, c], f)
    pt = (.7, -.4, 1.3)
    for mod in (m, m2):
        np.testing.assert_allclose(eval_exprs(mod._ode, pt[:2], pt[2:], []), np.asarray(F(*pt), dtype=float).ravel(), rtol=1e-14)
        Je = expr.jacobian(mod._ode, list(mod.x) + list(mod.u))
        np.testing.assert_allclose(eval_exprs([e for r in Je for e in r], pt[:2], pt[2:], []).reshape(2, 3), J(*pt), rtol=1e-12)
        g, fn, Jd, _, _ = derivative_dag(2, 1, mod._ode)
        np.testing.assert_allclose(np.array(g.evaluate([Jd[i][j] for i in range(2) for j in range(3)], pt[:2], pt[2:], [])).reshape(2, 3),
                                   J(*pt), rtol=1e-12)
    assert math.isclose(eval_exprs([expr.tan(expr.Expr.wrap(.3))], [], [], [])[0], math.tan(.3), rel_tol=1e-15)


def test_the_reference_function_table_values_and_derivatives():
    """util/parsing.py:36-58: log10 sign abs min max arcsin arccos arctan arctan2 arsinh arcosh artanh (with the functions already
    there: the whole table) - as text and as expressions; values, the expression-level derivative and the symbolic derivative DAG
    (first and contracted second derivatives) against sympy / finite differences; the non-smooth ones as CasADi differentiates
    them (d|a| = sign(a) da, d sign = 0, min / max: the active branch)."""
    import numpy as np
    import sympy as sp
    from hilo_mpc_amd import Model, expr
    from hilo_mpc_amd.parsing import FUNCTIONS
    from hilo_mpc_amd.symdiff import derivative_dag
    from tests.problems import eval_exprs
    assert sorted(FUNCTIONS) == sorted(['sqrt', 'exp', 'log', 'log10', 'sign', 'abs', 'min', 'max', 'sin', 'cos', 'tan', 'arcsin',
                                        'arccos', 'arctan', 'arctan2', 'sinh', 'cosh', 'tanh', 'arsinh', 'arcosh', 'artanh'])
    m = Model(name='table')
    m.set_dynamical_states(['a', 'b']), m.set_inputs(['c'])
    m.set_dynamical_equations(['log10(2 + a*a) + arcsin(a/3) * arccos(b/4) + arctan(a*b) + abs(a - b) * sign(c)',
                               'arctan2(a, 1 + c*c) + arsinh(b*c) + arcosh(2 + a*a) + artanh(b/5) + min(a*c, b) + max(a, b*b)'])
    a, b, c = sp.symbols('a b c', real=True)
    f = sp.Matrix([sp.log(2 + a * a, 10) + sp.asin(a / 3) * sp.acos(b / 4) + sp.atan(a * b) + sp.Abs(a - b) * sp.sign(c),
                   sp.atan2(a, 1 + c * c) + sp.asinh(b * c) + sp.acosh(2 + a * a) + sp.atanh(b / 5) + sp.Min(a * c, b) + sp.Max(a, b * b)])
    for pt in ((.7, -.4, 1.3), (-1.1, .9, -.6), (.2, .3, .8)):
        F = np.array(f.subs({a: pt[0], b: pt[1], c: pt[2]}).evalf(20), dtype=float).ravel()
        J = np.array(f.jacobian([a, b, c]).subs({a: pt[0], b: pt[1], c: pt[2]}).evalf(20), dtype=float)
        np.testing.assert_allclose(eval_exprs(m._ode, pt[:2], pt[2:], []), F, rtol=1e-14)
        Je = expr.jacobian(m._ode, list(m.x) + list(m.u))
        np.testing.assert_allclose(eval_exprs([e for r in Je for e in r], pt[:2], pt[2:], []).reshape(2, 3), J, rtol=1e-12, atol=1e-14)
        g, fn, Jd, H, kb = derivative_dag(2, 1, m._ode)
        np.testing.assert_allclose(np.array(g.evaluate([Jd[i][j] for i in range(2) for j in range(3)], pt[:2], pt[2:], [])).reshape(2, 3),
                                   J, rtol=1e-12, atol=1e-14)
        # contracted Hessian sum_m kb_m d2 f_m / dw2 (packed lower triangle) against sympy
        kbv = [.3, -1.7]
        Hs = kbv[0] * sp.hessian(f[0], [a, b, c]) + kbv[1] * sp.hessian(f[1], [a, b, c])
        Hn = np.array(Hs.subs({a: pt[0], b: pt[1], c: pt[2]}).evalf(20), dtype=float)
        got = g.evaluate(H, pt[:2], pt[2:], [], kb=kbv)
        np.testing.assert_allclose(got, [Hn[i][j] for i in range(3) for j in range(i + 1)], rtol=1e-11, atol=1e-13)
