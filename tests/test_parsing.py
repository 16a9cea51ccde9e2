"""Model equations given as text (`Model.set_equations(equations=...)`, dynamic_model.py:1508-1553; grammar of
util/parsing.py:246-545): declaration of variables by the way they appear, auxiliary definitions, constants, continuation
lines - against the same models built from expressions, and the reference's own example strings (tests/test_PFs.py:17-22)."""
import numpy as np
import pytest

from hilo_mpc_amd import Model
from tests.problems import eval_exprs, symbolic_model


def _same(a, b, x, u, p):
    np.testing.assert_allclose(eval_exprs(a, x, u, p), eval_exprs(b, x, u, p), rtol=1e-14)


def test_reference_example_linear_model():
    """tests/test_PFs.py:17-28: two states, one input, two parameters, one measurement - and the model is linear."""
    m = Model(name='lin')
    m.set_equations(equations='''
    dx_1/dt = -k_1*x_1(t) + u(k)
    dx_2/dt = k_1*x_1(t) - k_2*x_2(t)
    y(k) = x_2(t)
    ''')
    assert m.dynamical_state_names == ['x_1', 'x_2'] and m.input_names == ['u'] and m.parameter_names == ['k_1', 'k_2']
    assert m.measurement_names == ['y'] and (m.n_x, m.n_u, m.n_p, m.n_y, m.n_z) == (2, 1, 2, 1, 0)
    x, u, p = [.3, -.2], [.7], [2., 5.]
    np.testing.assert_allclose(eval_exprs(m._ode, x, u, p), [-2. * .3 + .7, 2. * .3 - 5. * -.2])
    np.testing.assert_allclose(eval_exprs(m._meas, x, u, p), [-.2])
    assert m.is_linear()
    m.discretize('erk', order=1, inplace=True)
    m.setup(dt=1.)
    from hilo_mpc_amd import PF, KF
    with pytest.warns(UserWarning, match="The supplied model is linear"):
        PF(m)
    KF(m)                                                        # accepted: a linear model written as text


def test_chemostat_as_text_equals_the_expression_model():
    ref = symbolic_model('chemostat4')
    m = Model(name='chemo_text')
    m.set_equations(equations=[
        '# growth with substrate inhibition',
        'phi = 0.407*S(t)/(0.108 + S(t) + S(t)^2/14814.0)',
        'mu = phi*(ISF + 0.22*IRF/(0.22 + I(t)))',
        'Rfp = phi*(0.0005 + I(t))/(0.022 + I(t))',
        'D = DS(k) + DI(k)',
        'dX/dt = mu*X(t) - D*X(t)',
        'd/dt(S(t)) = -(2.0*mu*X(t)) - D*S(t) ...',
        '             + DS(k)*Sf',
        'dP/dt = Rfp*X(t) - D*P(t)',
        'dI/dt = -(D*I(t)) + DI(k)*If',
        'yX(k) = X(t)',
        'yP(k) = P(t)',
        'X | description: biomass',
    ])
    assert m.dynamical_state_names == ['X', 'S', 'P', 'I'] and m.input_names == ['DS', 'DI']
    assert sorted(m.parameter_names) == sorted(['ISF', 'IRF', 'Sf', 'If']) and m.measurement_names == ['yX', 'yP']
    assert m._equation_notes == {'X': {'description': 'biomass'}} and not m.is_linear()
    rng = np.random.default_rng(0)
    for _ in range(3):
        x = np.array([.1, 40., .5, .2]) * (1 + .2 * rng.uniform(-1, 1, 4))
        u = rng.uniform(0, .3, 2)
        pv = dict(Sf=100., If=4., ISF=1., IRF=.3)
        _same(m._ode + m._meas, ref._ode + ref._meas, x, u, [pv[n] for n in m.parameter_names]) if m.parameter_names == ref.parameter_names else \
            np.testing.assert_allclose(eval_exprs(m._ode + m._meas, x, u, [pv[n] for n in m.parameter_names]),
                                       eval_exprs(ref._ode + ref._meas, x, u, [pv[n] for n in ref.parameter_names]), rtol=1e-13)


def test_discrete_model_with_dt_and_constants():
    """tests/test_PFs.py:39-44 pattern: a difference equation that uses the sampling interval; the emitted functor gets its value."""
    m = Model(name='toy_text', discrete=True)
    m.set_equations(equations="""
        a = 25
        x(k+1) = x(k)/2 + a*dt*x(k)/(1 + x(k)^2)
        y(k) = x(k)^2/20
    """)
    assert m.dynamical_state_names == ['x'] and m.measurement_names == ['y'] and m.n_p == 0 and m.discrete
    m.setup(dt=.5)
    src = m.user_source()
    assert 'DISCRETE = true' in src and 'dt' not in src.split('ode(')[1].split('}')[0].replace('double dt', '').replace('(void)dt', '')
    # same functor as the string form of set_dynamical_equations with dt (dynamic_model.py:1293)
    m2 = Model(name='toy_text2', discrete=True)
    m2.set_dynamical_states('x')
    m2.set_dynamical_equations('x/2 + 25*dt*x/(1 + x^2)')
    m2.set_measurement_equations('x^2/20')
    m2.setup(dt=.5)
    from hilo_mpc_amd.expr import Expr
    sub = lambda es: Expr.substitute(es, lambda n: Expr.wrap(.5) if n.op == 'dt' else None)
    _same(sub(m._ode + m._meas), sub(m2._ode + m2._meas), [1.7], [], [])
    np.testing.assert_allclose(eval_exprs(sub(m._ode), [1.7], [], []), [1.7 / 2 + 25 * .5 * 1.7 / (1 + 1.7 ** 2)])


def test_algebraic_states_in_both_forms():
    for alg in ('0 = h + l*cos(theta(t)) - yt(t)', 'yt(t) = h + l*cos(theta(t))'):
        m = Model(name='dae_text')
        m.set_equations(equations=['h = 0.5', 'l = 1.0', 'dtheta/dt = omega(t)', 'domega/dt = -9.81*sin(theta(t)) + F(k) + 0.1*yt(t)', alg,
                                   'y1(k) = theta(t)'])
        assert m.dynamical_state_names == ['theta', 'omega'] and m.algebraic_state_names == ['yt'] and m.input_names == ['F']
        assert m.n_z == 1 and len(m._alg) == 1 and m.parameter_names == []
        from hilo_mpc_amd.expr import Expr
        z = .5 + np.cos(.3)
        full = Expr.substitute(m._alg, lambda n: Expr.wrap(z) if n.op == 'z' else None)
        assert abs(eval_exprs(full, [.3, 0.], [0.], [])[0]) < 1e-15


def test_errors():
    m = Model(name='bad')
    with pytest.raises(NotImplementedError, match="quadrature"):
        m.set_equations(equations='int = x(t)^2')
    with pytest.raises(ValueError, match="refers to itself"):
        Model(name='bad2').set_equations(equations=['a = b + 1', 'b = a*2', 'dx/dt = a*x(t)'])
    with pytest.raises(TypeError):
        Model(name='bad3').set_equations(equations=3)
    with pytest.raises(RuntimeError, match="device zoo"):
        Model('chemostat4').set_equations(equations='dx/dt = x(t)')


def test_text_models_compile_for_the_device():
    """The functors emitted for text-defined models compile for gfx950 (hiprtc, no GPU): a continuous one with the symbolic
    derivative code for the tracking policy, a discrete one with `dt` for the filter kernels."""
    from hilo_mpc_amd import _lib
    m = Model(name='lin')
    m.set_equations(equations=['dx_1/dt = -k_1*x_1(t) + u(k)', 'dx_2/dt = k_1*x_1(t) - k_2*x_2(t)*x_1(t)', 'y(k) = x_2(t)'])
    m.discretize('rk4', inplace=True)
    m.setup(dt=.1)
    src = m.user_source()
    assert 'ModelSym<UserModel>' in src
    _lib.check(_lib.lib().hilo_jit_precompile(src.encode(), 0, 0, 0, 0, 0, 8, 0, 0, 0, 0, 0))
    d = Model(name='toy_text', discrete=True)
    d.set_equations(equations=['x(k+1) = x(k)/2 + 25*dt*x(k)/(1 + x(k)^2)', 'y(k) = x(k)^2/20'])
    d.setup(dt=1.)
    _lib.check(_lib.lib().hilo_jit_precompile_kf(d.user_source().encode()))


def test_reference_style_declaration_with_named_measurement():
    """tests/test_PFs.py:39-44 as written there."""
    m = Model(name='toy_ref', discrete=True)
    m.set_dynamical_states('x')
    m.set_measurements('y')
    m.set_dynamical_equations('x/2 + 25*dt*x/(1 + x^2)')
    m.set_measurement_equations('x^2/20')
    m.setup(dt=1.)
    assert m.measurement_names == ['y'] and m.n_y == 1 and not m.is_linear()
    from hilo_mpc_amd import PF
    pf = PF(m)
    assert pf.sample_size == 15 and pf.variant is None
