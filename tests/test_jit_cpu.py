"""Generated sources compile for gfx950 (hiprtc needs no GPU): the expression front-end (hilo_mpc_amd/codegen.py) against the
engine headers, including a learned term inside a model written as expressions.  Nothing is executed
here - the GPU suite (tests/test_jit_gpu.py, tests/test_hybrid_gpu.py) runs the same sources."""
import pytest

from hilo_mpc_amd import _lib, codegen
from tests.problems import symbolic_model


class _TrainedGp:
    """What `substitute_from` looks at: labels, features, a callable predict, a handle (never dereferenced here)."""

    def __init__(self, features, labels):
        self.features, self.labels, self._handle = list(features), list(labels), object()

    def predict(self, X):
        raise AssertionError


def _compile(src, policy=0, nth=0, ne=0, nc=0, coll_d=0, N=8, hold=0, cont=0, tv=0, big=0, has_fun=0):
    _lib.check(_lib.lib().hilo_jit_precompile(src.encode(), policy, nth, ne, nc, coll_d, N, hold, cont, tv, big, has_fun))


def test_learned_term_substitution_rewrites_the_parameter_vector():
    m = symbolic_model('chemostat4_mu')
    m.substitute_from(_TrainedGp(['S', 'DS', 'ISF'], ['mu']))
    assert m.parameter_names == ['Sf', 'If', 'ISF', 'IRF'] and m.n_p == 4 and len(m._gps) == 1
    src = m.user_source()
    assert 'gp_se_mean(hilo_user_gp[0], g' in src and 'p[4]' not in src
    # the features in the GP's order: state S = x[1], input DS = u[0], parameter ISF = p[2]
    line = [ln for ln in src.splitlines() if '_t(' in ln][0]
    assert [q.split('(')[1].rstrip(')') for q in line.split('{')[1].split('}')[0].split(', ')] == ['x[1]', 'u[0]', 'p[2]']


def test_substitute_errors():
    m = symbolic_model('chemostat4_mu')
    with pytest.raises(ValueError, match="not a parameter"):
        m.substitute_from(_TrainedGp(['S'], ['X']))
    with pytest.raises(ValueError, match="is not a state, input or parameter"):
        m.substitute_from(_TrainedGp(['nope'], ['mu']))
    with pytest.raises(ValueError, match="exactly one label"):
        m.substitute_from(_TrainedGp(['S'], ['mu', 'Sf']))
    untrained = _TrainedGp(['S'], ['mu'])
    untrained._handle = None
    with pytest.raises(RuntimeError, match="has not been set up"):
        m.substitute_from(untrained)


@pytest.mark.parametrize('policy', [0, 2])
def test_generated_hybrid_model_compiles(policy):
    m = symbolic_model('chemostat4_mu')
    m.substitute_from(_TrainedGp(['S', 'I'], ['mu']))
    src = m.user_source()
    if policy == 2:
        src += codegen.fun_source(m.n_x)
    _compile(src, policy=policy, has_fun=int(policy == 2))


def test_compile_error_is_reported_with_the_compiler_log():
    with pytest.raises(ValueError, match="run-time compilation"):
        _compile("struct UserModel { this is not C++ };\n")


@pytest.mark.parametrize('name', ['pendulum4_dae', 'chemostat4_dae'])
def test_generated_dae_model_compiles_with_the_collocation_policy(name):
    """Semi-explicit DAE (set_algebraic_states / set_algebraic_equations): the emitted model solves its algebraic equations
    inside `ode` in whatever scalar type it is called with; general policy, Radau-3 collocation, continuous objective."""
    m = symbolic_model(name)
    assert m.n_z == 1
    src = m.user_source(z_guess=[1.4])
    assert 'NZ = 1' in src and 'dae_ode<UserModel>' in src and 'alg_jz' in src
    _compile(src + codegen.fun_source(m.n_x), policy=2, coll_d=3, cont=1, has_fun=1)


@pytest.mark.parametrize('name', ['chemostat4', 'pendulum4', 'pendulum4_dae'])
def test_filter_kernels_of_an_expression_model_compile(name):
    """KF / EKF / UKF on a model written as expressions: the six kernels kf_body<UserModel, UKF, MODE> (predict / update / step)."""
    m = symbolic_model(name)
    _lib.check(_lib.lib().hilo_jit_precompile_kf(m.user_source().encode()))


def _table_model():
    """Every function of the reference's table (util/parsing.py:36-58) in one continuous model, written as text."""
    from hilo_mpc_amd import Model
    m = Model(name='table')
    m.set_equations(equations='''
    da/dt = -a(t) + log10(2 + a(t)^2) + arcsin(a(t)/30) * arccos(b(t)/40) + arctan(a(t)*b(t)) + abs(a(t) - b(t)) * sign(c(k))
    db/dt = -b(t) + arctan2(a(t), 1 + c(k)^2) + arsinh(b(t)*c(k)) + arcosh(2 + a(t)^2) + artanh(b(t)/50) ...
            + min(a(t)*c(k), b(t)) + max(a(t), b(t)^2) + tanh(a(t)) + sqrt(1 + exp(-b(t)))
    y(k) = a(t) + abs(b(t))
    ''')
    return m


@pytest.mark.parametrize('policy', [0, 2])
def test_model_with_the_whole_function_table_compiles(policy):
    """The emitted functor is evaluated in every scalar type of the engine: plain / fast doubles, second-order Taylor numbers
    (general policy, and the tracking policy's values) and - through the generated symbolic derivatives - straight-line code
    (tracking policy): all of them need every function of the table (csrc/hilo_ad.h)."""
    m = _table_model()
    assert (m.n_x, m.n_u, m.n_y) == (2, 1, 1)
    src = m.user_source()
    for name in ('log10(', 'asin(', 'acos(', 'atan(', 'atan2(', 'asinh(', 'acosh(', 'atanh(', 'fabs(', 'sign('):
        assert name in src, name
    if policy == 2:
        src += codegen.fun_source(m.n_x)
    _compile(src, policy=policy, has_fun=int(policy == 2))


def test_filter_kernels_with_the_whole_function_table_compile():
    """... and first-order dual numbers (Kalman filters)."""
    _lib.check(_lib.lib().hilo_jit_precompile_kf(_table_model().user_source().encode()))


@pytest.mark.parametrize('coll_d', [0, 3])
@pytest.mark.parametrize('name', ['chemostat4', 'table'])
def test_moving_horizon_estimator_policy_compiles_for_expression_models(name, coll_d):
    """JIT_MHE (csrc/hilo_mhe_policy.h around the emitted functor): explicit Runge-Kutta with symbolic derivatives, and the
    reference's default transcription, collocation (Taylor path), for a zoo twin and for a model without one."""
    m = symbolic_model(name) if name != 'table' else _table_model()
    _compile(m.user_source(), policy=3, coll_d=coll_d, N=6)


def test_moving_horizon_estimator_policy_compiles_for_a_zoo_functor_under_collocation():
    from hilo_mpc_amd import Model
    _compile(Model('chemostat4').user_source(), policy=3, coll_d=2, N=5)
