"""GPU parity of the stochastic NMPC: `SMPC` of the reference (mpc.py:2462-2808) on the device against the
oracle's restatement (oracle/smpc.py), through the committed fixture tests/golden/smpc.json (made by
tests/golden/make_smpc_golden.py from the oracle, so that no GPU time is spent in sympy).  The surrogate model - posterior mean,
its derivative and the posterior variance of the learned term, the written-out Runge-Kutta step and its Jacobian, the covariance
propagation - runs inside the run-time compiled general policy."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.problems import SMPC_CASES, smpc_models, smpc_product, smpc_product_gp      # noqa: E402

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'smpc.json')))


def _gp(name):
    m, _ = smpc_models(name)
    return smpc_product_gp(m.dynamical_state_names[SMPC_CASES[name]['features'][0]])


@pytest.mark.parametrize('name', ['siso', 'mimo', 'pend'])
def test_surrogate_map_on_the_device_vs_oracle(name):
    """x+ of the compiled surrogate (means and column-major covariance entries) at the fixture's random points: gp_se_mean /
    gp_se_dmean / gp_se_var of csrc/hilo_models.h against the oracle GP written out in sympy."""
    smpc = smpc_product(name, _gp(name))
    assert smpc._jit and 'gp_se_var(hilo_user_gp[0]' in smpc._user_source
    pts = GOLD[name]['points']
    xa, u, p, f = (np.array([q[k] for q in pts]) for k in ('xa', 'u', 'p', 'f'))
    got = smpc.plant_step(xa, u, cp=p).cpu().numpy()
    np.testing.assert_allclose(got, f, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize('name', ['siso', 'pend'])
def test_smpc_solve_vs_oracle(name):
    """Batch of four starts: same status, decision vector (means, covariance entries, inputs) and cost as the oracle; default
    tolerance and a tight one (the fixed point itself)."""
    c = SMPC_CASES[name]
    g = GOLD[name]
    x0 = np.array(g['x0'])
    gp = _gp(name)
    for tag, opts, tol_v in (('default', {}, 5e-5), ('tight', {'tol': 1e-10}, 1e-6)):
        ref = g['solves'][tag]
        smpc = smpc_product(name, gp, **opts)
        assert (smpc._n_v, smpc._n_g) == (g['n_v'], g['n_g'])
        u = smpc.optimize(x0, cov_x0=c['cov0'], Kgain=c['K'])
        st = smpc.solver_status_code
        assert np.array_equal(st, ref['status']) and np.all(st == 1), (st, ref['status'])
        v, vr = smpc._nlp_solution['x'].cpu().numpy(), np.array(ref['v'])
        assert np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))) < tol_v, tag
        np.testing.assert_allclose(smpc._nlp_solution['f'].cpu().numpy(), ref['f'], rtol=1e-7)
        np.testing.assert_allclose(u, np.array(ref['u0']), rtol=1e-4, atol=1e-6)
    # the gain given to the constructor instead of optimize(): the same problem without gain parameters
    fixed = smpc_product(name, gp, Kgain=np.asarray(c['K']), tol=1e-10)
    assert fixed._n_p == 0
    uf = fixed.optimize(x0, cov_x0=c['cov0'])
    np.testing.assert_allclose(uf, u, rtol=1e-7, atol=1e-9)
    # single instance: (n_u x 1) like the reference's DM; covariance prediction grows along the horizon
    u1 = smpc.optimize(x0[0], cov_x0=c['cov0'], Kgain=c['K'])
    assert u1.shape == (len(c['K']), 1)
    np.testing.assert_allclose(u1[:, 0], u[0], rtol=1e-6, atol=1e-8)


def test_reference_smoke_configurations():
    """tests/test_SMPC.py:104-110 and :160-170 as written there (no input weights, gain 0, zero initial covariance, GP
    hyper-parameters from `fit_model`): the calls go through and end with a solver status of the reference's table."""
    from hilo_mpc_amd import GP, SMPC
    from tests.problems import smpc_training_data
    X, y = smpc_training_data()
    for name, x_lb, p_lb, cov0, K in (('siso', [10], .9, [0], 0), ('mimo', [0, 0], [.97, .97], np.zeros((2, 2)), np.zeros((2, 2)))):
        m, _ = smpc_models(name)
        gp = GP([m.dynamical_state_names[0]], 'z', solver='ipopt')
        gp.set_training_data(X, y)
        gp.setup()
        gp.fit_model()
        smpc = SMPC(m, gp, np.asarray(SMPC_CASES[name]['Bw']))
        smpc.horizon = 10
        names = m.dynamical_state_names
        smpc.quad_stage_cost.add_states(names=names, ref=[1] * len(names), weights=[10] * len(names))
        smpc.quad_terminal_cost.add_states(names=names, ref=[1] * len(names), weights=[10] * len(names))
        smpc.set_box_chance_constraints(x_lb=x_lb, x_lb_p=p_lb)
        smpc.setup(options={'chance_constraints': 'prs', 'print_level': 0})
        u = smpc.optimize(x0=SMPC_CASES[name]['x0'], cov_x0=cov0, Kgain=K)
        assert u.shape == (m.n_u, 1) and smpc.solver_status_code[0] in (1, 2, 3, 4, 5, -1)
        xp, up = smpc.return_prediction()[:2]
        assert xp.shape == (1, m.n_x + m.n_x ** 2, 11) and up.shape == (1, m.n_u, 10)
        if smpc.solver_status_code[0] in (1, 2):
            n = m.n_x                                   # predicted variances grow from the zero start, one GP variance per step
            assert np.all(np.diff(xp[0, n]) > 0) and xp[0, n, 0] == 0.


def test_posterior_variance_inside_the_compiled_model_at_200_training_points():
    """The learned term of BASELINE configuration 4 has 200 training points (mpc.py:2573-2612 propagates its posterior variance): the
    surrogate's compiled `gp_se_var` / `gp_se_mean` against `GaussianProcess.predict` (the prediction kernel, pinned by the
    reference's GP known answers) on the single integrator with zero state covariance and zero gain, where the propagated
    covariance IS the posterior variance at the mean state and the mean map is x + u + mean(x)."""
    from hilo_mpc_amd import GP, Kernel
    rng = np.random.default_rng(8)
    X = np.linspace(0., 3., 200)[None, :]
    y = np.sin(2. * X) + .05 * rng.standard_normal(X.shape)
    gp = GP(['px'], ['z'], kernel=Kernel.squared_exponential(active_dims=[0], length_scales=[.5], signal_variance=1.), noise_variance=1e-2)
    gp.set_training_data(X, y)
    gp.setup()
    smpc = smpc_product('siso', gp)
    assert 'gp_se_var(hilo_user_gp[0]' in smpc._user_source
    xq = rng.uniform(.2, 2.8, 16)
    u = rng.uniform(-.3, .3, (16, 1))
    xa = np.stack([xq, np.zeros(16)], axis=1)                     # [mean | covariance entry]
    got = smpc.plant_step(xa, u, cp=np.zeros((16, 1))).cpu().numpy()
    mean, var = gp.predict(xq[None, :])
    mean, var = np.asarray(mean).reshape(-1), np.asarray(var).reshape(-1)
    np.testing.assert_allclose(got[:, 0], xq + u[:, 0] + mean, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(got[:, 1], var, rtol=1e-7, atol=1e-10)
