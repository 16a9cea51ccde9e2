"""GPU parity of the estimator variants the reference's own MHE tests configure (tests/test_MHE.py:20-110, :150-230, :331; row a11):
NO state noise, parameters estimated through `quad_arrival_cost.add_parameters` on models written as expressions, under the
default collocation or with a discretised model - the general policy csrc/hilo_mhe_policy.h::MheGen against the oracle's
simultaneous form (oracle/mhe_gen.py, which reproduces oracle/mhe.py and oracle/mhe_coll.py on their cases).
Tolerances at a matched tight tolerance (tol 1e-10): v 1e-6 relative, objective 1e-9, multipliers 1e-5."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import models                                              # noqa: E402
from oracle.mhe_gen import MheGenIpm, MheGenProblem                    # noqa: E402
from oracle.nmpc import IpmOptions                                     # noqa: E402
from tests.problems import C3B, c3_data, symbolic_model                # noqa: E402
from tests.util import golden_or_compute                               # noqa: E402

TOL = 1e-10
P_TRUE = [100., 4., 1., 0.]


def _product(spec, options, est=None, noise=True, symbolic=True):
    """est = dict(p_lb, p_ub, Wp, p_guess[, p_scaling]) over ALL model parameters (the reference's `add_parameters`, modeling.py:762-777)"""
    from hilo_mpc_amd import MHE, Model
    m = symbolic_model(spec['model']) if symbolic else Model(spec['model'])
    if options.get('integration_method') == 'discrete':
        m = m.discretize('erk', order=spec.get('order', 4))
    m = m.setup(dt=spec['dt'])
    mhe = MHE(m)
    mhe.quad_arrival_cost.add_states(weights=list(spec['Wx']), guess=spec['x_guess'])
    mhe.quad_stage_cost.add_measurements(weights=list(spec['Wy']))
    if noise:
        mhe.quad_stage_cost.add_state_noise(weights=list(spec['Ww']))
    mhe.horizon = spec['N']
    if est:
        mhe.quad_arrival_cost.add_parameters(weights=est['Wp'], guess=est['p_guess'])
        mhe.set_box_constraints(x_lb=spec.get('x_lb'), x_ub=spec.get('x_ub'), w_lb=spec.get('w_lb') if noise else None,
                                w_ub=spec.get('w_ub') if noise else None, p_lb=est['p_lb'], p_ub=est['p_ub'])
        mhe.set_initial_guess(x_guess=spec['x_guess'], p_guess=est['p_guess'])
        if est.get('p_scaling') is not None:
            mhe.set_scaling(p_scaling=est['p_scaling'])
    else:
        mhe.set_box_constraints(x_lb=spec.get('x_lb'), x_ub=spec.get('x_ub'), w_lb=spec.get('w_lb') if noise else None,
                                w_ub=spec.get('w_ub') if noise else None, p_lb=spec['p'], p_ub=spec['p'])
        mhe.set_initial_guess(x_guess=spec['x_guess'])
    mhe.setup(options=options, nlp_opts={'ipopt.tol': TOL})
    return mhe


def _oracle(spec, degree, noise, est_idx=(), **kw):
    k = {q: v for q, v in spec.items() if q not in ('model', 'p', 'order') and (noise or q not in ('Ww', 'w_lb', 'w_ub'))}
    pb = MheGenProblem(models.get(spec['model']), degree=degree, order=spec.get('order', 4), noise=noise, est=list(est_idx), **k, **kw)
    return pb, MheGenIpm(pb, IpmOptions(tol=TOL))


def _pb_data(pb):
    """what _compare reads of an oracle problem, as plain data (fixtures)"""
    if isinstance(pb, dict):
        return pb
    return dict(n_v=int(pb.n_v), n_g=int(pb.n_g), x_ind=pb.x_ind, w_ind=pb.w_ind, ip_ind=pb.ip_ind, p_ind=pb.p_ind)


def _compare(mhe, pb, ref, x_opt, p_opt, B, vtol=1e-6):
    pb = _pb_data(pb)
    plain = lambda v: json.loads(json.dumps(v))                                        # noqa: E731  (tuples / numpy ints -> lists / ints)
    assert (mhe._n_v, mhe._n_g) == (pb['n_v'], pb['n_g'])
    assert plain(mhe._x_ind) == plain(pb['x_ind']) and plain(mhe._w_ind) == plain(pb['w_ind'])
    assert plain(mhe._ip_ind) == plain(pb['ip_ind']) and plain(mhe._p_ind) == plain(pb['p_ind'])
    assert np.array_equal(mhe.solver_status_code, ref['status']) and np.all(ref['status'] == 1)
    v, vr = mhe._nlp_solution['x'].cpu().numpy(), ref['v']
    assert np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))) < vtol
    np.testing.assert_allclose(mhe._nlp_solution['f'].cpu().numpy(), ref['f'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(x_opt.cpu().numpy(), ref['x_opt'], rtol=1e-6, atol=1e-8)
    lam, lr = mhe._nlp_solution['lam_g'].cpu().numpy(), ref['lam']
    assert np.max(np.abs(lam - lr) / np.maximum(1., np.abs(lr))) < 1e-5


def param_est_case(method, noise):
    """Oracle side of test_parameter_estimation_on_an_expression_model (minutes of sympy for the discretised model: kept as a
    fixture under tests/golden/, made by tests/golden/make_mhe_gen_fixtures.py from THIS function)."""
    N, B = 8, 4
    spec = dict(C3B, N=N)
    xa, um, ym, _ = c3_data(B, N=N)
    degree = 3 if method == 'collocation' else 0
    pb, ipm = _oracle(spec, degree, noise, est_idx=[2], Wp=[1e-2], p_lb=[.2], p_ub=[2.], p_guess=[.7])
    ref = ipm.solve(xa, [.7], [100., 4., 0.], um, ym)
    y_new = ym[:, -1] * 1.001
    um2, ym2 = np.concatenate([um[:, 1:], um[:, -1:]], axis=1), np.concatenate([ym[:, 1:], y_new[:, None]], axis=1)
    ref2 = ipm.solve(ref['X'][:, 2] * pb.sx, ref['P'], [100., 4., 0.], um2, ym2, w0=ref['w'])
    keep = ('status', 'v', 'f', 'x_opt', 'lam', 'p_opt')
    return dict(pb=_pb_data(pb), ref={k: np.asarray(ref[k]) for k in keep}, ref2={k: np.asarray(ref2[k]) for k in ('status', 'x_opt', 'p_opt')})


def hard_con_case(method, noise):
    """Oracle side of test_hard_stage_constraint_vs_oracle."""
    N, B = 5, 4
    spec = dict(C3B, N=N)
    xa, um, ym, _ = c3_data(B, N=N)
    degree = 3 if method == 'collocation' else 0
    _, free = _oracle(spec, degree, noise)
    sol_free = free.solve(xa, [], P_TRUE, um, ym)
    x_free = sol_free['X']
    ub = float(np.round(x_free[:, :N, 0].max(axis=1).min() * .97, 4))                    # active in every instance, at a node k < N
    cons = dict(expr=['X', 'P + 2*I*X'], lb=[-np.inf, 0.], ub=[ub, np.inf])
    pb, ipm = _oracle(spec, degree, noise, constraint=cons)
    ref = ipm.solve(xa, [], P_TRUE, um, ym)
    ref['lam'] = ipm.lam_g(ref)
    top = ref['X'][:, :N, 0].max(axis=1) if not degree else np.maximum(ref['X'][:, :N, 0].max(axis=1), ref['Xc'][..., 0].max(axis=(1, 2)))
    assert np.all(top > ub - 1e-7) and np.all(top < ub + 1e-7) and np.all(ref['f'] > sol_free['f'] * 1.1)
    return dict(pb=_pb_data(pb), ub=ub, ref={k: np.asarray(ref[k]) for k in ('status', 'v', 'f', 'x_opt', 'lam')})


EST = dict(p_lb=[100., 4., .2, 0.], p_ub=[100., 4., 2., 0.], Wp=np.diag([0., 0., 1e-2, 0.]), p_guess=[100., 4., .7, 0.])


@pytest.mark.parametrize('method,noise', [('collocation', False), ('collocation', True), ('discrete', False), ('discrete', True)])
def test_parameter_estimation_on_an_expression_model(method, noise):
    """chemostat4 written as expressions, ISF estimated from a wrong arrival value (0.7, truth 1.0), the other three parameters
    pinned by p_lb == p_ub; with and without state noise; collocation (Radau 3, the reference's default) and the discretised model."""
    N, B = 8, 4
    spec = dict(C3B, N=N)
    xa, um, ym, _ = c3_data(B, N=N)
    y_new = ym[:, -1] * 1.001
    R = golden_or_compute(f'mhe_gen_param_est_{method}_{int(noise)}', lambda: param_est_case(method, noise))
    ref, ref2 = R['ref'], R['ref2']
    mhe = _product(spec, {'integration_method': method}, est=EST, noise=noise)
    assert mhe._estimating and mhe.has_state_noise == noise
    for k in range(N):
        mhe.add_measurements(ym[:, k], um[:, k])
    x_opt, p_opt = mhe.estimate(x_arrival=xa, p_arrival=[100., 4., .7, 0.])
    _compare(mhe, R['pb'], ref, x_opt, p_opt, B)
    np.testing.assert_allclose(p_opt.cpu().numpy()[:, 2], ref['p_opt'][:, 0], rtol=1e-6)
    np.testing.assert_allclose(p_opt.cpu().numpy()[:, [0, 1, 3]], np.tile([100., 4., 0.], (B, 1)), rtol=0, atol=0)
    assert np.all(np.abs(ref['p_opt'][:, 0] - 1.) < .3)                                # the estimate moved from 0.7 towards the truth
    if not noise:
        assert mhe.return_mhe_estimation()[1] is None
    # next sample: window shifts, arrival values from the previous solution (smoothing, mhe.py:254-256), warm start
    mhe.add_measurements(y_new, um[:, -1])
    x2, p2 = mhe.estimate()
    assert np.array_equal(mhe.solver_status_code, ref2['status'])
    np.testing.assert_allclose(x2.cpu().numpy(), ref2['x_opt'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(p2.cpu().numpy()[:, 2], ref2['p_opt'][:, 0], rtol=1e-5)


def test_no_state_noise_with_pinned_parameters_and_scaling():
    """Without state noise and without estimated parameters (tests/test_MHE.py:20-60 with the parameters pinned): n_x degrees of
    freedom; x and p scaling; degree 2 Legendre points."""
    N, B = 6, 4
    spec = dict(C3B, N=N, x_scaling=[1., 10., 1., 1.])
    xa, um, ym, _ = c3_data(B, N=N)
    pb, ipm = _oracle(spec, 2, False, points='legendre')
    ref = ipm.solve(xa, [], P_TRUE, um, ym)
    from hilo_mpc_amd import MHE
    m = symbolic_model('chemostat4').setup(dt=spec['dt'])
    mhe = MHE(m)
    mhe.quad_arrival_cost.add_states(weights=list(spec['Wx']), guess=spec['x_guess'])
    mhe.quad_stage_cost.add_measurements(weights=list(spec['Wy']))
    mhe.horizon = N
    mhe.set_box_constraints(x_lb=spec['x_lb'], p_lb=P_TRUE, p_ub=P_TRUE)
    mhe.set_initial_guess(x_guess=spec['x_guess'])
    mhe.set_scaling(x_scaling=spec['x_scaling'])
    mhe.setup(options={'integration_method': 'collocation', 'degree': 2, 'collocation_points': 'legendre'}, nlp_opts={'ipopt.tol': TOL})
    assert not mhe.has_state_noise and mhe._n_v == 4 + 7 * 4 + 6 * 8
    for k in range(N):
        mhe.add_measurements(ym[:, k], um[:, k])
    x_opt, p_opt = mhe.estimate(x_arrival=xa)
    _compare(mhe, pb, ref, x_opt, p_opt, B)
    np.testing.assert_allclose(p_opt.cpu().numpy(), np.tile(P_TRUE, (B, 1)))


def test_reference_test_2_configuration_runs():
    """tests/test_MHE.py:62-110 as it is written: dX = 3 - X - k1 X, one parameter estimated between 0 and 5, no state noise, default
    options (collocation), horizon 10 - estimate() returns (None, None) until the window is full, then states and parameter close to
    the truth (the reference's test only prints them)."""
    from hilo_mpc_amd import MHE, Model
    m = Model(name='mhe_test_2')
    x = m.set_dynamical_states(['x0'])
    p = m.set_parameters(['k1'])
    m.set_measurements(['y1'])
    m.set_measurement_equations([x[0]])
    m.set_dynamical_equations([3.0 - x[0] - p[0] * x[0]])
    m.setup(dt=.5)
    mhe = MHE(m)
    mhe.horizon = 10
    mhe.quad_arrival_cost.add_states(weights=[10], guess=[5.])
    mhe.quad_arrival_cost.add_parameters(weights=[10], guess=[1])
    mhe.quad_stage_cost.add_measurements(weights=[10])
    mhe.set_box_constraints(x_lb=[0], x_ub=[6], p_lb=[0], p_ub=[5])
    mhe.setup()
    xs, X = [], 5.
    for k in range(12):                                # truth with k1 = 1: x' = 3 - 2 x
        xs.append(X)
        X = 1.5 + (X - 1.5) * np.exp(-2 * .5)
    out = None
    for k in range(12):
        mhe.add_measurements(np.array([[xs[k]]]))
        out = mhe.estimate()
        assert (out == (None, None)) == (k < 9)
    x_est, p_est = out
    assert mhe.solver_status_code[0] == 1
    # (not exact although the data are: the 'smoothing' update hands x_2 of the previous window to a window that starts one sample
    # later, mhe.py:254-256, and the arrival cost pulls x_0 - and with it k1 - towards that)
    assert abs(float(p_est[0, 0]) - 1.) < 5e-2 and abs(float(x_est[0, 0]) - X) < 5e-2


@pytest.mark.parametrize('method,noise', [('collocation', True), ('collocation', False), ('discrete', True), ('discrete', False)])
def test_hard_stage_constraint_vs_oracle(method, noise):
    """`mhe.stage_constraint` (mhe.py:498-508; the reference's own use: tests/test_MHE.py:605-660, a band on a concentration under
    collocation with state noise): hard rows at every collocation point (:536-553) and at every node k < N (:749-757), on the NLP's
    SCALED variables.  Two expressions, one of them nonlinear; the upper bound on the biomass is 97 % of the unconstrained
    estimate's maximum, so rows are active; the multipliers come back in the reference's row order."""
    N, B = 5, 4
    spec = dict(C3B, N=N)
    xa, um, ym, _ = c3_data(B, N=N)
    degree = 3 if method == 'collocation' else 0
    R = golden_or_compute(f'mhe_gen_hard_con_{method}_{int(noise)}', lambda: hard_con_case(method, noise))
    ref, ub = R['ref'], float(R['ub'])
    from hilo_mpc_amd import MHE
    m = symbolic_model('chemostat4')
    if method == 'discrete':
        m = m.discretize('erk', order=4)
    m = m.setup(dt=spec['dt'])
    mhe = MHE(m)
    mhe.quad_arrival_cost.add_states(weights=list(spec['Wx']), guess=spec['x_guess'])
    mhe.quad_stage_cost.add_measurements(weights=list(spec['Wy']))
    if noise:
        mhe.quad_stage_cost.add_state_noise(weights=list(spec['Ww']))
    mhe.horizon = N
    mhe.set_box_constraints(x_lb=spec.get('x_lb'), x_ub=spec.get('x_ub'), w_lb=spec.get('w_lb') if noise else None,
                            w_ub=spec.get('w_ub') if noise else None, p_lb=P_TRUE, p_ub=P_TRUE)
    mhe.set_initial_guess(x_guess=spec['x_guess'])
    x = m.x
    mhe.stage_constraint.constraint = [x[0], x[2] + 2 * x[3] * x[0]]
    mhe.stage_constraint.lb = [-np.inf, 0.]
    mhe.stage_constraint.ub = [ub, np.inf]
    mhe.setup(options={'integration_method': method}, nlp_opts={'ipopt.tol': TOL})
    for k in range(N):
        mhe.add_measurements(ym[:, k], um[:, k])
    x_opt, p_opt = mhe.estimate(x_arrival=xa)
    # (collocation with noise: one noise variable sits at its bound 1e-3 with a vanishing multiplier - no strict complementarity,
    # its distance from the bound goes with sqrt(mu) and differs by 3e-6 between two runs that stop one iteration apart)
    _compare(mhe, R['pb'], ref, x_opt, p_opt, B, vtol=1e-5 if (degree and noise) else 1e-6)
    lam = mhe._nlp_solution['lam_g'].cpu().numpy().reshape(B, N, -1)
    d, nx = degree, 4
    rows = np.concatenate([lam[:, :, :2 * d], lam[:, :, 2 * d + d * nx + nx:]], axis=2)
    assert np.abs(rows).max() > 1. and np.abs(rows[:, :, 1::2]).max() < 1e-8             # the active bound carries force; expression 2 never binds


def test_soft_stage_constraint_is_refused_like_the_reference_fails():
    from hilo_mpc_amd import MHE
    m = symbolic_model('chemostat4').setup(dt=C3B['dt'])
    mhe = MHE(m)
    mhe.quad_arrival_cost.add_states(weights=list(C3B['Wx']), guess=C3B['x_guess'])
    mhe.quad_stage_cost.add_measurements(weights=list(C3B['Wy']))
    mhe.quad_stage_cost.add_state_noise(weights=list(C3B['Ww']))
    mhe.horizon = 4
    mhe.stage_constraint.constraint = [m.x[0]]
    mhe.stage_constraint.ub = [1.]
    mhe.stage_constraint.is_soft = True
    with pytest.raises(NotImplementedError, match='soft'):
        mhe.setup()
