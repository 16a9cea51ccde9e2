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


def test_substitute_from_errors():
    from hilo_mpc_amd import GP, Kernel, Model
    gp = GP(['S', 'I'], ['mu'])
    with pytest.raises(RuntimeError, match="has not been set up"):
        Model('chemostat4').substitute_from(gp)
    with pytest.raises(NotImplementedError, match="no learnable term"):
        Model('pendulum4').substitute_from(gp)
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
