"""Stochastic NMPC on the CPU: the expression-level derivatives against sympy, the product's surrogate model
against the oracle's restatement of `SMPC._create_deterministic_surrogate` (mpc.py:2512-2614) at random points, the error
behaviour of the reference's own tests (tests/test_SMPC.py), and that the whole problem - surrogate with learned-term nodes,
chance-constraint rows, trace cost - compiles for gfx950 (hiprtc needs no GPU).  The GPU suite (tests/test_smpc_gpu.py)
solves the same problems."""
import numpy as np
import pytest
import sympy as sp

from hilo_mpc_amd import SMPC, expr
from hilo_mpc_amd.smpc import _erfinv
from tests.problems import (SMPC_CASES, eval_exprs, smpc_models, smpc_oracle_post, smpc_oracle_problem,
                            symbolic_model)


class _TrainedGp:
    """What the surrogate builder looks at: features, labels, the training inputs' shape, a handle (never dereferenced here)."""

    def __init__(self, features, labels=('z',), n=6):
        self.features, self.labels, self._handle = list(features), list(labels), object()
        self.X_train = np.zeros((len(self.features), n))


def _oracle_gp_functions(post):
    from oracle.smpc import gp_symbolic
    f = sp.Symbol('f0')
    mean, var = gp_symbolic(post, [f])
    fm, fv, fd = sp.lambdify([f], mean), sp.lambdify([f], var), sp.lambdify([f], sp.diff(mean, f))
    return dict(mean=lambda a: float(fm(a[0])), var=lambda a: float(fv(a[0])), dmean=lambda a, j: float(fd(a[0])))


def test_erfinv_matches_scipy():
    from scipy.special import erfinv
    for y in [-.999999, -.9, -.5, -1e-3, 0., 1e-8, .3, .8, .908, .954, .99, .999999999]:
        assert abs(_erfinv(y) - erfinv(y)) <= 2e-15 * max(1., abs(erfinv(y))), y
    assert _erfinv(1.) == np.inf and _erfinv(-1.) == -np.inf


@pytest.mark.parametrize('name', ['chemostat4', 'pendulum4', 'cstr3'])
def test_expression_jacobian_matches_sympy(name):
    """`expr.jacobian` (the `ca.jacobian` that becomes part of the surrogate MODEL) against the oracle's sympy Jacobians."""
    from oracle import models
    m, om = symbolic_model(name), models.get(name)
    J = expr.jacobian(m._ode, list(m.x) + list(m.u))
    rng = np.random.default_rng(3)
    base = {'chemostat4': ([.1, 40., .5, .2], [.1, .2], [100., 4., 1., 0.]), 'pendulum4': ([.1, .2, .3, .4], [.5], []),
            'cstr3': ([.6, .4, 430.], [1e4], [])}[name]
    for _ in range(5):
        x, u = [np.asarray(b, dtype=float) * (1 + .2 * rng.uniform(-1, 1, len(b))) for b in base[:2]]
        p = np.asarray(base[2], dtype=float)
        got = eval_exprs([e for row in J for e in row], x, u, p).reshape(m.n_x, m.n_x + m.n_u)
        ref = np.concatenate([om.fx(x, u, p, 1.)[0], om.fu(x, u, p, 1.)[0]], axis=1)
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-14 * np.abs(ref).max())


@pytest.mark.parametrize('name,fixed_gain', [('siso', False), ('mimo', False), ('mimo', True), ('pend', False), ('pend', True)])
def test_surrogate_matches_the_oracle(name, fixed_gain):
    """Mean and covariance propagation of the product's surrogate (expression trees with gp / gpd / gpvar nodes, Runge-Kutta step
    written out) against the oracle's sympy surrogate at random means, covariances, inputs and gains."""
    from oracle.smpc import smpc_surrogate
    c = SMPC_CASES[name]
    m, om = smpc_models(name)
    post = smpc_oracle_post()
    K = np.asarray(c['K'], dtype=float)
    smpc = SMPC(m, _TrainedGp([m.dynamical_state_names[c['features'][0]]]), np.asarray(c['Bw']), Kgain=K if fixed_gain else None)
    sur, _, _ = smpc_surrogate(om, [post], [c['features']], c['Bw'], K if fixed_gain else None)
    mc = smpc._model
    n, nu = m.n_x, m.n_u
    assert mc.n_x == n + n * n == sur.nx and mc.n_u == nu and mc.n_p == (0 if fixed_gain else n * nu) == sur.np_
    assert mc.dynamical_state_names == m.dynamical_state_names + [f'kx_{k}' for k in range(n * n)]
    assert mc.discrete and mc.dt == 1.
    gpf = [_oracle_gp_functions(post)]
    rng = np.random.default_rng(11)
    for _ in range(6):
        mean = np.asarray(c['x0']) * (1 + .3 * rng.uniform(-1, 1, n)) + rng.uniform(-.5, .5, n)
        if name != 'pend':
            mean[0] = rng.uniform(-.5, 1.5)                 # inside the training data, where the learned term is not flat
        A = rng.uniform(-1, 1, (n, n))
        cov = .1 * A @ A.T                                  # symmetric positive semi-definite
        u = rng.uniform(-1, 1, nu)
        p = np.zeros(0) if fixed_gain else (K * (1 + .5 * rng.uniform(-1, 1, K.shape))).T.reshape(-1)   # column-major
        xa = np.concatenate([mean, cov.T.reshape(-1)])
        got = eval_exprs(mc._ode, xa, u, p, gpf)
        ref = sur.f(xa, u, p, 1.)[0]
        np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12)
        # the propagated covariance is symmetric (column-major vec)
        Kn = got[n:].reshape(n, n).T
        np.testing.assert_allclose(Kn, Kn.T, rtol=1e-10, atol=1e-13)


def test_covariance_propagation_against_the_formula_evaluated_numerically():
    """The oracle's own restatement against plain numpy: Kx+ = [J B] bigK [J B]^T with finite-difference Jacobians of the
    discretised plant and of the oracle GP's `predict` (independent of both symbolic builds)."""
    from oracle.smpc import smpc_surrogate
    c = SMPC_CASES['pend']
    _, om = smpc_models('pend')
    post = smpc_oracle_post()
    K = np.asarray(c['K'], dtype=float)
    sur, _, _ = smpc_surrogate(om, [post], [c['features']], c['Bw'], K)
    rng = np.random.default_rng(2)
    mean, u = np.array([.4, -.2]), np.array([.3])
    A = rng.uniform(-1, 1, (2, 2))
    cov = .05 * A @ A.T
    Bw = np.asarray(c['Bw'])

    def jac(fun, z, h=1e-6):
        return np.stack([(fun(z + h * e) - fun(z - h * e)) / (2 * h) for e in np.eye(len(z))], axis=1)

    z = np.concatenate([mean, u])
    J = jac(lambda q: om.f(q[:2], q[2:], [], 1.)[0], z)
    jgp = jac(lambda q: post.predict(q[[0]][:, None])[0][0], z)
    mu, var = [float(v[0, 0]) for v in post.predict(mean[[0]][:, None])]
    Kz = np.block([[cov, cov @ K.T], [K @ cov, K @ cov @ K.T]])
    Kd = np.array([[var]]) + jgp @ Kz @ jgp.T
    Kzd = Kz @ jgp.T
    bigK = np.block([[Kz, Kzd], [Kzd.T, Kd]])
    JB = np.concatenate([J, Bw], axis=1)
    ref_mean = om.f(mean, u, [], 1.)[0] + Bw[:, 0] * mu
    ref_cov = JB @ bigK @ JB.T
    got = sur.f(np.concatenate([mean, cov.T.reshape(-1)]), u, [], 1.)[0]
    np.testing.assert_allclose(got[:2], ref_mean, rtol=1e-12)
    np.testing.assert_allclose(got[2:].reshape(2, 2).T, ref_cov, rtol=1e-6, atol=1e-9)


def test_oracle_solves_the_reference_test_system():
    """tests/test_SMPC.py:104-110 made regular: the chance constraint is active at the end of the horizon - the mean keeps
    sqrt(2) erfinv(2 p - 1) standard deviations away from the bound - and the covariance grows by the GP variance per step."""
    from oracle.nmpc_gen import GenIpm
    from scipy.special import erfinv
    pb = smpc_oracle_problem('siso')
    c = SMPC_CASES['siso']
    ipm = GenIpm(pb)
    r = ipm.solve(np.array([c['x0'] + [0.]]), np.array([[0.]]))
    assert r['status'][0] == 1
    v = ipm.to_v(r)[0]
    X = v[:2 * (c['N'] + 1)].reshape(-1, 2)
    far = 1. + 1e-2                                          # far from the data: prior variance + noise variance
    np.testing.assert_allclose(np.diff(X[:, 1]), far, rtol=1e-6)
    margin = X[:, 0] - np.sqrt(2.) * erfinv(2 * .9 - 1) * np.sqrt(X[:, 1] + 1e-8) - 10.
    assert np.all(margin > -1e-6) and abs(margin[-1]) < 1e-5


# ---- the reference's interface tests (tests/test_SMPC.py:86-118) ------------------------------------------------------------------
def _smpc(name='siso'):
    m, _ = smpc_models(name)
    return SMPC(m, _TrainedGp([m.dynamical_state_names[0]]), np.asarray(SMPC_CASES[name]['Bw']))


def test_box_constraints():
    smpc = _smpc()
    smpc.set_box_chance_constraints(x_lb=[10])
    assert smpc._x_lb == [10., 0.] and smpc._x_lb_p == [.954] and smpc.x_lb_s == [10.]
    smpc = _smpc('mimo')
    smpc.set_box_chance_constraints(x_lb=[-100, 0], x_ub=[100, 30], x_lb_p=[.95, .95])
    inf = float('inf')
    assert smpc._x_lb == [-100., 0., 0., -inf, -inf, 0.] and smpc._x_ub == [100., 30., inf, inf, inf, inf]


def test_box_constraints_1():
    with pytest.raises(TypeError, match="probabilities must be between 0 and 1"):
        _smpc().set_box_chance_constraints(x_lb=[10], x_lb_p=2)
    with pytest.raises(TypeError, match="Use 'set_box_chance_constraints' instead"):
        _smpc().set_box_constraints(x_lb=[10])
    for f in ('set_stage_constraints', 'set_terminal_constraints', 'set_custom_constraints_function'):
        with pytest.raises(NotImplementedError):
            getattr(_smpc(), f)()


def test_not_passing_k0():
    smpc = _smpc()
    with pytest.raises(ValueError, match="cov_x0"):
        smpc.optimize(x0=[15.], Kgain=0)


def test_surrogate_errors():
    from hilo_mpc_amd import Model
    m, _ = smpc_models('siso')
    with pytest.raises(NotImplementedError, match="written as expressions"):
        SMPC(Model('chemostat4').discretize('rk4').setup(dt=1.), _TrainedGp(['S']), np.ones((4, 1)))
    with pytest.raises(ValueError, match="is not in list"):
        SMPC(m, _TrainedGp(['nope']), [[1.]])
    g = _TrainedGp(['px'])
    g._handle = None
    with pytest.raises(RuntimeError, match="has not been set up"):
        SMPC(m, g, [[1.]])
    with pytest.raises(NotImplementedError, match="256 training points"):      # (csrc/hilo_models.h::GP_VAR_MAX; configuration 4 has 200)
        SMPC(m, _TrainedGp(['px'], n=257), [[1.]])
    with pytest.raises(ValueError, match="B must have shape"):
        SMPC(m, _TrainedGp(['px']), [[1., 2.]])


@pytest.mark.parametrize('name,fixed_gain', [('siso', False), ('mimo', False), ('pend', True)])
def test_whole_problem_compiles(name, fixed_gain, monkeypatch):
    """`setup()` with HILO_JIT_COMPILE_ONLY=1: surrogate model (learned-term nodes, written-out Runge-Kutta step), chance
    constraint rows as stage and terminal constraints and the trace cost, compiled for gfx950 against the engine headers."""
    from tests.problems import smpc_product
    monkeypatch.setenv('HILO_JIT_COMPILE_ONLY', '1')
    c = SMPC_CASES[name]
    m, _ = smpc_models(name)
    smpc = smpc_product(name, _TrainedGp([m.dynamical_state_names[0]]), Kgain=np.asarray(c['K']) if fixed_gain else None)
    src = smpc._user_source
    assert 'gp_se_mean(hilo_user_gp[0]' in src and 'gp_se_var(hilo_user_gp[0]' in src and 'gp_se_dmean(hilo_user_gp[0]' in src
    assert 'struct UserFun' in src and 'NEXPR = %d' % (2 * m.n_x) in src and 'NTEXPR = %d' % (2 * m.n_x) in src
    assert 'ModelSym<UserModel>' not in src                       # learned terms keep the Taylor sweeps
    assert not smpc._nlp_setup_done
    with pytest.raises(ValueError, match="need to setup"):
        smpc.optimize(c['x0'], cov_x0=c['cov0'], Kgain=c['K'])


@pytest.mark.parametrize('name', ['siso', 'pend'])
def test_oracle_smpc_solution_vs_slsqp(name):
    """Independent solver on the same transcription (scipy SLSQP on the reference's NLP in its own variables, x_0 substituted):
    same minimiser and cost as the oracle's interior point - the cross-check the other unpinned NMPC oracles get."""
    from oracle.nmpc_gen import GenIpm
    from tests.test_oracle_nmpc_gen import _slsqp
    c = SMPC_CASES[name]
    pb = smpc_oracle_problem(name)
    ipm = GenIpm(pb)
    n = len(c['x0'])
    xa0 = np.concatenate([np.asarray(c['x0'], dtype=float), np.asarray(c['cov0'], dtype=float).T.reshape(-1)])[None]
    p = np.asarray(c['K'], dtype=float).T.reshape(1, -1)
    res = ipm.solve(xa0, p)
    assert res['status'][0] == 1 and res['kkt'][0] <= 1e-8
    _slsqp(pb, ipm, res, 0, p[0], strict=False)


def test_surrogate_with_two_learned_terms():
    """Two GPs on different states (`SMPC(model, [gp_1, gp_2], B)` with B of shape n_x x 2): stacked means, block-diagonal
    variances and stacked Jacobians of the surrogate against the oracle."""
    from oracle.smpc import smpc_surrogate
    m, om = smpc_models('pend')
    post = smpc_oracle_post()
    Bw = np.array([[.05, 0.], [.02, .1]])
    K = np.array([[-.5, -.3]])
    smpc = SMPC(m, [_TrainedGp(['th']), _TrainedGp(['om'])], Bw, Kgain=K)
    sur, _, _ = smpc_surrogate(om, [post, post], [[0], [1]], Bw, K)
    gpf = [_oracle_gp_functions(post)] * 2
    rng = np.random.default_rng(5)
    for _ in range(4):
        mean = rng.uniform(-.3, 1.2, 2)
        A = rng.uniform(-1, 1, (2, 2))
        cov = .05 * A @ A.T
        u = rng.uniform(-1, 1, 1)
        xa = np.concatenate([mean, cov.T.reshape(-1)])
        np.testing.assert_allclose(eval_exprs(smpc._model._ode, xa, u, [], gpf), sur.f(xa, u, [], 1.)[0], rtol=1e-10, atol=1e-12)
    assert len(smpc._model._gps) == 2
