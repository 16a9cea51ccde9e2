"""GPU parity of the GP-hybrid NMPC: `Model.substitute_from(gp)` puts the posterior mean
of a trained GaussianProcess into the right-hand side of the chemostat; the oracle writes the same mean out term by
term in sympy and solves the transcription with its dense interior-point method."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.nmpc import DenseIpm                                                  # noqa: E402
from tests.problems import C2, C4, c2_x0, oracle_c4, product_gp, product_nmpc    # noqa: E402


@pytest.fixture(scope='module')
def oracle():
    pb, post = oracle_c4()
    return pb, post, DenseIpm(pb)


def test_gp_mean_inside_the_model_matches_predict(oracle):
    """x+ of the hybrid shooting map vs the oracle's, and the oracle's unrolled mean vs Posterior.predict."""
    pb, post, _ = oracle
    rng = np.random.default_rng(5)
    B = 64
    x = np.array([.1, 40., 0., 0.]) * (1 + .3 * rng.uniform(-1, 1, (B, 4))) + np.array([0, 0, .5, .5]) * rng.uniform(0, 1, (B, 4))
    u = rng.uniform(0, 1, (B, 2))
    nmpc = product_nmpc(C4)
    xn = nmpc.plant_step(x, u, cp=C4['p']).cpu().numpy()
    xr = pb.phi(x / pb.sx, u / pb.su, C4['p']) * pb.sx
    np.testing.assert_allclose(xn, xr, rtol=1e-11, atol=1e-13)
    # the mean the model sees is gp.predict(...)[0] (dynamic_model.py:3065-3071)
    gp = product_gp()
    mean, _ = gp.predict(x[:, [1, 3]].T)
    mref, _ = post.predict(x[:, [1, 3]].T)
    np.testing.assert_allclose(np.asarray(mean), mref, rtol=1e-9, atol=1e-11)


def test_c4_solve_vs_oracle(oracle):
    pb, _, ipm = oracle
    B = 8
    x0 = c2_x0(B)
    ref = ipm.solve(x0, C4['p'])
    nmpc = product_nmpc(C4)
    u = nmpc.optimize(x0, cp=C4['p'])
    st = nmpc.solver_status_code
    assert np.array_equal(st, ref['status']) and np.all(st == 1), (ref['status'], st)     # status parity, no masking
    ok = st == 1
    v = nmpc._nlp_solution['x'].cpu().numpy()
    vr = ipm.to_v(ref)
    scale = np.maximum(1., np.abs(vr))
    assert np.max((np.abs(v - vr) / scale)[ok]) < 5e-5
    np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy()[ok], ref['f'][ok], rtol=1e-7)
    np.testing.assert_allclose(u[ok], ref['u0'][ok], rtol=1e-4, atol=1e-6)


def test_golden_c4_closed_loop():
    """Committed fixture (tests/golden/make_hybrid_golden.py): GP trained on the fixture's own data, then the closed
    loop of tests/test_nmpc_gpu.py's fixture check (status exact, v 5e-5 at tol 1e-8, 2e-6 at tol 1e-9)."""
    import json
    import os
    from tests.test_nmpc_gpu import GOLD, _check_against_fixture
    fx = json.load(open(os.path.join(GOLD, 'nmpc_c4.json')))
    gp = product_gp(np.array(fx['gp']['X']), np.array(fx['gp']['y']))
    assert abs(gp.log_marginal_likelihood() - fx['gp']['lml']) < 1e-7 * abs(fx['gp']['lml'])
    _check_against_fixture(lambda tol: product_nmpc(C4, gp=gp, **({'tol': tol} if tol else {})), fx, fx['p'])


def test_hybrid_differs_from_first_principles_and_closes_the_loop():
    """The learned growth rate changes the optimum (the GP is not a no-op), and a warm-started closed loop at the
    C4 batch size per GPU (256) converges everywhere."""
    B = 256
    x0 = c2_x0(B)
    hyb, fp = product_nmpc(C4), product_nmpc(C2)
    uh, uf = hyb.optimize(x0, cp=C4['p']), fp.optimize(x0, cp=C2['p'])
    assert np.abs(uh - uf).max() > 1e-3
    x = x0
    for _ in range(5):
        u = hyb.optimize(x, cp=C4['p'])
        x = hyb.plant_step(x, u, cp=C4['p']).cpu().numpy()
    assert np.mean(hyb.solver_status_code == 1) >= 0.99
    assert np.all(np.isfinite(x)) and np.all(x >= -1e-6)


def test_learned_term_in_a_model_written_as_expressions_equals_the_zoo_hybrid():
    """`substitute_from` as a mechanism (row a17): the chemostat written as expressions with a PARAMETER `mu`, replaced by
    the trained GP and compiled at setup, against the precompiled hybrid functor - same shooting map (the kernel sum is
    taken in a different order: 1e-12), same optimum and status."""
    from tests.problems import symbolic_model
    gp = product_gp()
    m = symbolic_model('chemostat4_mu')
    assert m.parameter_names == ['Sf', 'If', 'ISF', 'IRF', 'mu']
    m.substitute_from(gp)
    assert m.parameter_names == ['Sf', 'If', 'ISF', 'IRF'] and m.n_p == 4          # the label leaves the parameter vector
    sym, zoo = product_nmpc(C4, model=m), product_nmpc(C4, gp=gp)
    assert sym._jit and 'gp_se_mean(hilo_user_gp[0]' in sym._user_source
    rng = np.random.default_rng(7)
    B = 64
    x = np.array([.1, 40., 0., 0.]) * (1 + .3 * rng.uniform(-1, 1, (B, 4))) + np.array([0, 0, .5, .5]) * rng.uniform(0, 1, (B, 4))
    u = rng.uniform(0, 1, (B, 2))
    np.testing.assert_allclose(sym.plant_step(x, u, cp=C4['p']).cpu().numpy(), zoo.plant_step(x, u, cp=C4['p']).cpu().numpy(),
                               rtol=1e-12, atol=1e-14)
    x0 = c2_x0(8)
    us, uz = sym.optimize(x0, cp=C4['p']), zoo.optimize(x0, cp=C4['p'])
    assert np.array_equal(sym.solver_status_code, zoo.solver_status_code) and np.all(zoo.solver_status_code == 1)
    np.testing.assert_allclose(us, uz, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(sym._nlp_solution['f'].cpu().numpy(), zoo._nlp_solution['f'].cpu().numpy(), rtol=1e-10)
    # a second controller on a DIFFERENT GP shares the code object but not the learned-term table
    X, y = __import__('tests.problems', fromlist=['c4_training_data']).c4_training_data(seed=99)
    gp2 = product_gp(X, 1.5 * y)
    m2 = symbolic_model('chemostat4_mu')
    m2.substitute_from(gp2)
    sym2 = product_nmpc(C4, model=m2)
    u2 = sym2.optimize(x0, cp=C4['p'])
    assert np.abs(u2 - us).max() > 1e-3
    # the first one is unaffected (warm-started now: same optimum to solver accuracy, not the same round-off path)
    np.testing.assert_allclose(sym.optimize(x0, cp=C4['p']), us, rtol=1e-6, atol=1e-8)


def test_learned_term_over_state_input_and_parameter_features():
    """Features are looked up by name among states, inputs and parameters (dynamic_model.py:3056-3061); three features,
    constant mean.  The compiled shooting map against RK4 in numpy with `gp.predict` for the rate."""
    from hilo_mpc_amd import GP, Kernel, Mean, NMPC
    from tests.problems import symbolic_model
    rng = np.random.default_rng(11)
    n = 60
    Xt = np.stack([rng.uniform(0, 40, n), rng.uniform(0, 1, n), rng.uniform(.5, 1.5, n)])
    yt = (0.4 * Xt[0] / (1. + Xt[0]) * Xt[2] - .05 * Xt[1])[None, :]
    gp = GP(['S', 'DS', 'ISF'], ['mu'], kernel=Kernel.squared_exponential(active_dims=[0, 1, 2], length_scales=[8., .7, .6],
                                                                          ard=True, signal_variance=.8),
            mean=Mean.constant(bias=.1), noise_variance=1e-3)
    gp.set_training_data(Xt, yt)
    gp.setup()
    m = symbolic_model('chemostat4_mu')
    m.substitute_from(gp)
    m = m.discretize('erk', order=4).setup(dt=.5)
    nmpc = NMPC(m)
    nmpc.quad_stage_cost.add_states(names=['X'], weights=[1.], ref=[1.])
    nmpc.quad_stage_cost.add_inputs(names=['DS', 'DI'], weights=[.1, .1])      # a regular problem: every input is priced
    nmpc.horizon = 5
    nmpc.setup(options={'integration_method': 'discrete'})
    B = 32
    x = np.array([.1, 30., .1, .2]) * (1 + .3 * rng.uniform(-1, 1, (B, 4)))
    u = rng.uniform(0, 1, (B, 2))
    p = np.array([40., 4., 1., .3])

    def rhs(x):
        X, S, Pr, I = x.T
        mu = np.asarray(gp.predict(np.stack([S, u[:, 0], np.full(B, p[2])]))[0]).ravel()
        phi = 0.407 * S / (0.108 + S + S * S / 14814.0)
        Rs = 2.0 * (phi * (p[2] + 0.22 * p[3] / (0.22 + I)))
        Rfp = phi * (0.0005 + I) / (0.022 + I)
        D = u[:, 0] + u[:, 1]
        return np.stack([mu * X - D * X, -(Rs * X) - D * S + u[:, 0] * p[0], Rfp * X - D * Pr, -(D * I) + u[:, 1] * p[1]], 1)

    h = .5
    k1 = rhs(x); k2 = rhs(x + h / 2 * k1); k3 = rhs(x + h / 2 * k2); k4 = rhs(x + h * k3)
    ref = x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    np.testing.assert_allclose(nmpc.plant_step(x, u, cp=p).cpu().numpy(), ref, rtol=1e-9, atol=1e-11)
    nmpc.optimize(x[:4], cp=p)
    assert np.all(nmpc.solver_status_code == 1)


@pytest.mark.parametrize('kind', ['matern_52', 'se_plus_rq'])
def test_learned_term_with_another_kernel_is_compiled_into_the_model(kind):
    """Round 3: a learned term whose kernel is not the plain squared exponential - the kernel is compiled into the model source
    (hilo_mpc_amd/gp.py::kernel_expr, codegen.py::gp_helper_source) and differentiated by the Taylor sweeps like the rest of the
    right-hand side.  The compiled shooting map against RK4 in numpy with `gp.predict` (parity-tested for every kernel family)
    for the rate; the solve converges (exact first and second derivatives are what makes it)."""
    from hilo_mpc_amd import GP, Kernel, Mean, NMPC
    from tests.problems import symbolic_model
    rng = np.random.default_rng(12)
    n = 50
    Xt = np.stack([rng.uniform(0, 40, n), rng.uniform(0, 2, n)])
    yt = (0.4 * Xt[0] / (1. + Xt[0]) * (1. + .22 / (.22 + Xt[1])))[None, :]
    if kind == 'matern_52':
        kern = Kernel.matern_52(active_dims=[0, 1], length_scales=[12., 1.], ard=True, signal_variance=.6)
    else:
        kern = Kernel.squared_exponential(active_dims=[0, 1], length_scales=[12., 1.], ard=True, signal_variance=.5) + \
            Kernel.rational_quadratic(active_dims=[0], length_scales=20., alpha=1.5, signal_variance=.1)
    gp = GP(['S', 'I'], ['mu'], kernel=kern, mean=Mean.constant(bias=.2), noise_variance=1e-4)
    gp.set_training_data(Xt, yt)
    gp.setup()
    m = symbolic_model('chemostat4_mu')
    m.substitute_from(gp)
    assert 'hilo_user_gpk0' in m.user_source()
    m = m.discretize('erk', order=4).setup(dt=.5)
    nmpc = NMPC(m)
    nmpc.quad_stage_cost.add_states(names=['X'], weights=[1.], ref=[1.])
    nmpc.quad_stage_cost.add_inputs(names=['DS', 'DI'], weights=[.1, .1])
    nmpc.horizon = 5
    nmpc.set_box_constraints(u_lb=[0., 0.], u_ub=[1., 1.])
    nmpc.setup(options={'integration_method': 'discrete'})
    B = 32
    x = np.array([.1, 30., .1, .2]) * (1 + .3 * rng.uniform(-1, 1, (B, 4)))
    u = rng.uniform(0, 1, (B, 2))
    p = np.array([40., 4., 1., .3])

    def rhs(x):
        X, S, Pr, I = x.T
        mu = np.asarray(gp.predict(np.stack([S, I]))[0]).ravel()
        phi = 0.407 * S / (0.108 + S + S * S / 14814.0)
        Rs = 2.0 * (phi * (p[2] + 0.22 * p[3] / (0.22 + I)))
        Rfp = phi * (0.0005 + I) / (0.022 + I)
        D = u[:, 0] + u[:, 1]
        return np.stack([mu * X - D * X, -(Rs * X) - D * S + u[:, 0] * p[0], Rfp * X - D * Pr, -(D * I) + u[:, 1] * p[1]], 1)

    h = .5
    k1 = rhs(x); k2 = rhs(x + h / 2 * k1); k3 = rhs(x + h / 2 * k2); k4 = rhs(x + h * k3)
    ref = x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    np.testing.assert_allclose(nmpc.plant_step(x, u, cp=p).cpu().numpy(), ref, rtol=1e-9, atol=1e-11)
    nmpc.optimize(x[:8], cp=p)
    assert np.all(nmpc.solver_status_code == 1) and np.all(nmpc.stats()['kkt_error'] <= 1e-8)
    assert np.all(nmpc.stats()['iter_count'] < 60)


def test_substitute_from_errors():
    from hilo_mpc_amd import GP, Kernel, Model
    gp = GP(['S', 'I'], ['mu'])
    with pytest.raises(RuntimeError, match="has not been set up"):
        Model('chemostat4').substitute_from(gp)
    with pytest.raises(NotImplementedError, match="no learnable term"):
        Model('pendulum4').substitute_from(product_gp())
    bad = GP(['X', 'S'], ['mu'])
    with pytest.raises(ValueError, match="labels"):
        Model('chemostat4').substitute_from(bad)
    # a kernel the model functor has no closed form for is refused at NMPC.setup, loudly
    from tests.problems import c4_training_data
    X, y = c4_training_data()
    g2 = GP(['S', 'I'], ['mu'], kernel=Kernel.matern_32(active_dims=[0, 1], length_scales=[10., 1.], ard=True), noise_variance=1e-4)
    g2.set_training_data(X, y)
    g2.setup()
    from hilo_mpc_amd import NMPC
    from hilo_mpc_amd._lib import HiloError
    m = Model('chemostat4')
    m.substitute_from(g2)
    m = m.discretize('rk4').setup(dt=1.)
    nmpc = NMPC(m)
    nmpc.quad_stage_cost.add_states(names=['P'], weights=[10.], ref=[2.])
    nmpc.horizon = 5
    with pytest.raises(HiloError, match="squared-exponential"):
        nmpc.setup(options={'integration_method': 'discrete'})
