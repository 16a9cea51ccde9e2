"""GPU parity of NMPC with the reference's DEFAULT transcription, direct collocation at Radau points.
The oracle carries the collocation states as decision variables exactly like the reference (oracle/nmpc_coll.py); the
device solves the same collocation equations inside its shooting map (csrc/hilo_colloc.h), so the two meet at the KKT point:
solution, objective, collocation states and multipliers are compared there (iterates and iteration counts are not)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import models                                            # noqa: E402
from oracle.nmpc_coll import CollIpm, CollNmpcProblem                # noqa: E402
from tests.problems import C2, c2_x0                                 # noqa: E402


def _product(spec, objective=None, **solver_options):
    """`objective` None = the reference's default for a continuous model, 'continuous' (optimizer.py:1423-1426): the problem is
    then compiled at run time on the general policy; 'discrete' uses the precompiled collocation variant."""
    from hilo_mpc_amd import NMPC, Model
    m = Model(spec['model']).setup(dt=spec['dt'])                    # continuous model: NMPC's default is collocation
    nmpc = NMPC(m)
    xs, us = m.dynamical_state_names, m.input_names
    for ind, W, ref in spec.get('stage_states', []):
        nmpc.quad_stage_cost.add_states(names=[xs[i] for i in ind], weights=list(W), ref=ref)
    for ind, W, ref in spec.get('stage_inputs', []):
        nmpc.quad_stage_cost.add_inputs(names=[us[i] for i in ind], weights=list(W), ref=ref)
    for ind, W, ref in spec.get('terminal_states', []):
        nmpc.quad_terminal_cost.add_states(names=[xs[i] for i in ind], weights=list(W), ref=ref)
    nmpc.horizon = spec['N']
    nmpc.set_box_constraints(x_ub=spec.get('x_ub'), x_lb=spec.get('x_lb'), u_ub=spec.get('u_ub'), u_lb=spec.get('u_lb'))
    nmpc.set_initial_guess(x_guess=spec.get('x_guess'), u_guess=spec.get('u_guess'))
    if spec.get('x_scaling') or spec.get('u_scaling'):
        nmpc.set_scaling(x_scaling=spec.get('x_scaling'), u_scaling=spec.get('u_scaling'))
    # options default: integration_method='collocation'
    nmpc.setup(options=None if objective is None else {'objective_function': objective}, solver_options=solver_options or None)
    return nmpc


def _oracle(spec, objective='continuous'):
    kw = {k: v for k, v in spec.items() if k not in ('model', 'p', 'order')}
    pb = CollNmpcProblem(models.get(spec['model']), objective=objective, **kw)
    return pb, CollIpm(pb)


@pytest.mark.parametrize('objective', ['discrete', 'continuous'])
@pytest.mark.parametrize('over', [{}, dict(x_scaling=[.1, 40., 2., 1.], u_scaling=[2., 2.])])
def test_c2_collocation_vs_oracle(over, objective):
    spec = dict(C2, N=10, **over)
    x0 = c2_x0(6)
    pb, ipm = _oracle(spec, objective)
    ref = ipm.solve(x0, spec['p'])
    ok = ref['status'] == 1        # the oracle's simplified restoration gives up on one start of the continuous-objective problem
    assert ok.sum() >= len(x0) - 1
    x0, ref = x0[ok], {k: (v[ok] if isinstance(v, np.ndarray) and v.shape[:1] == ok.shape else v) for k, v in ref.items()}
    nmpc = _product(spec, None if objective == 'continuous' else objective)
    assert nmpc._nlp_options['objective_function'] == objective and nmpc._jit == (objective == 'continuous')
    assert nmpc._nlp_options['integration_method'] == 'collocation' and nmpc._nlp_options['degree'] == 3
    assert (nmpc._n_v, nmpc._n_g) == (pb.n_v, pb.n_g) == (11 * 4 + 10 * 2 + 10 * 12, 10 * 16)   # SURVEY 8a row a1 pattern
    assert nmpc._ip_ind == pb.ip_ind
    u = nmpc.optimize(x0, cp=spec['p'])
    assert np.array_equal(nmpc.solver_status_code, ref['status'])
    v, vr = nmpc._nlp_solution['x'].cpu().numpy(), ipm.to_v(ref)
    assert np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))) < 5e-5          # incl. the collocation states
    np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy(), ref['f'], rtol=1e-8)
    np.testing.assert_allclose(u, ref['u0'], rtol=5e-5, atol=1e-6)
    # multipliers in the reference's g order: per stage [collocation rows | continuity]
    lam = ref['lam'].reshape(len(x0), pb.N, -1).copy()
    lam[:, -1, -pb.nx:] += 2 * (ref['X'][:, -1] - pb.xrefN) @ pb.WN                # terminal cost on the end state (mpc.py:1682)
    got = nmpc._nlp_solution['lam_g'].cpu().numpy().reshape(len(x0), pb.N, -1)
    np.testing.assert_allclose(got[:, :, -pb.nx:], lam[:, :, -pb.nx:], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(got[:, :-1, :-pb.nx], lam[:, :-1, :-pb.nx], rtol=2e-4, atol=2e-5)
    # closed loop: warm start from the full previous vector (collocation states included), un-shifted
    x1 = c2_x0(6, seed=3)[ok]
    ref2 = ipm.solve(x1, spec['p'], w0=ref['w'])
    u2 = nmpc.optimize(x1, cp=spec['p'])
    assert np.array_equal(nmpc.solver_status_code, ref2['status'])
    np.testing.assert_allclose(u2, ref2['u0'], rtol=5e-5, atol=1e-6)


def test_legendre_points_vs_oracle():
    """options={'collocation_points': 'legendre'} (optimizer.py:1410-1418): Gauss points - only the basis the host passes
    changes (modeling.py:1091-1127); the end point is then extrapolated (D_0 != 0)."""
    spec = dict(C2, N=8)
    x0 = c2_x0(4)
    kw = {k: v for k, v in spec.items() if k not in ('model', 'p', 'order')}
    pb = CollNmpcProblem(models.get(spec['model']), points='legendre', objective='continuous', **kw)
    ipm = CollIpm(pb)
    ref = ipm.solve(x0, spec['p'])
    assert np.all(ref['status'] == 1) and abs(pb.D[0]) > 1e-3
    from hilo_mpc_amd import NMPC, Model
    m = Model(spec['model']).setup(dt=spec['dt'])
    nmpc = NMPC(m)
    xs, us = m.dynamical_state_names, m.input_names
    nmpc.set_quadratic_stage_cost(states=[xs[i] for i in spec['stage_states'][0][0]], cost_states=list(spec['stage_states'][0][1]),
                                  states_references=spec['stage_states'][0][2],
                                  inputs=[us[i] for i in spec['stage_inputs'][0][0]], cost_inputs=list(spec['stage_inputs'][0][1]))
    nmpc.set_quadratic_terminal_cost(states=[xs[i] for i in spec['terminal_states'][0][0]],
                                     cost=list(spec['terminal_states'][0][1]), references=spec['terminal_states'][0][2])
    nmpc.horizon = spec['N']
    nmpc.set_box_constraints(x_lb=spec.get('x_lb'), u_ub=spec.get('u_ub'), u_lb=spec.get('u_lb'))
    nmpc.set_initial_guess(x_guess=spec.get('x_guess'), u_guess=spec.get('u_guess'))
    nmpc.setup(options={'collocation_points': 'legendre'})
    u = nmpc.optimize(x0, cp=spec['p'])
    assert np.array_equal(nmpc.solver_status_code, ref['status'])
    v, vr = nmpc._nlp_solution['x'].cpu().numpy(), ipm.to_v(ref)
    assert np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))) < 5e-5
    np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy(), ref['f'], rtol=1e-8)
    np.testing.assert_allclose(u, ref['u0'], rtol=5e-5, atol=1e-6)
    radau = _product(spec)
    ur = radau.optimize(x0, cp=spec['p'])
    assert np.abs(ur - u).max() > 1e-7                                  # a different discretisation, not the same numbers


@pytest.mark.parametrize('objective', ['discrete', 'continuous'])
@pytest.mark.parametrize('degree,points', [(1, 'radau'), (2, 'radau'), (2, 'legendre'), (4, 'radau')])
def test_other_degrees_vs_oracle(degree, points, objective):
    """options={'degree': d}: the same implicit shooting map with d collocation points (modeling.py:1091-1211)."""
    spec = dict(C2, N=6)
    x0 = c2_x0(3)
    kw = {k: v for k, v in spec.items() if k not in ('model', 'p', 'order')}
    pb = CollNmpcProblem(models.get(spec['model']), degree=degree, points=points, objective=objective, **kw)
    ipm = CollIpm(pb)
    ref = ipm.solve(x0, spec['p'])
    assert np.all(ref['status'] == 1)
    from hilo_mpc_amd import NMPC, Model
    m = Model(spec['model']).setup(dt=spec['dt'])
    nmpc = NMPC(m)
    nmpc.quad_stage_cost.add_states(names=['P'], weights=[10.], ref=[2.])
    nmpc.quad_stage_cost.add_inputs(names=m.input_names, weights=[.1, .1])
    nmpc.quad_terminal_cost.add_states(names=['P'], weights=[10.], ref=[2.])
    nmpc.horizon = spec['N']
    nmpc.set_box_constraints(x_lb=spec.get('x_lb'), u_ub=spec.get('u_ub'), u_lb=spec.get('u_lb'))
    nmpc.set_initial_guess(x_guess=spec.get('x_guess'), u_guess=spec.get('u_guess'))
    nmpc.setup(options={'degree': degree, 'collocation_points': points, 'objective_function': objective})
    assert (nmpc._n_v, nmpc._n_g) == (pb.n_v, pb.n_g) and nmpc._ip_ind == pb.ip_ind
    u = nmpc.optimize(x0, cp=spec['p'])
    assert np.array_equal(nmpc.solver_status_code, ref['status'])
    v, vr = nmpc._nlp_solution['x'].cpu().numpy(), ipm.to_v(ref)
    assert np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))) < 5e-5
    np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy(), ref['f'], rtol=1e-8)
    np.testing.assert_allclose(u, ref['u0'], rtol=5e-5, atol=1e-6)


def test_collocation_differs_from_rk4_and_satisfies_its_equations():
    """Not RK4 in disguise: the collocation optimum differs from the ERK-4 one, and the returned collocation states satisfy
    dt f(x_ki, u_k) = sum_j C[j,i] x_kj and x_{k+1} = sum_j D_j x_kj to round-off."""
    from tests.problems import product_nmpc
    spec = dict(C2, N=10)
    x0 = c2_x0(4)
    nmpc, rk = _product(spec), product_nmpc(spec)
    uc, ur = nmpc.optimize(x0, cp=spec['p']), rk.optimize(x0, cp=spec['p'])
    assert 1e-5 < np.abs(uc - ur).max() < 5e-2
    pb, ipm = _oracle(spec)
    v = nmpc._nlp_solution['x'].cpu().numpy()
    w = ipm.w_from_v(v)
    _, c = ipm.eval_fc(w, {'x0': x0 / pb.sx, 'p': np.tile(spec['p'], (4, 1))})
    cc = np.abs(c).reshape(4, pb.N, 4, 4)
    assert cc[:, :, :3].max() < 1e-11 and cc[:, :, 3].max() <= 1e-8      # collocation rows: round-off; continuity: the NLP tolerance


def test_integration_method_validation():
    from hilo_mpc_amd import NMPC, Model
    m = Model('chemostat4').setup(dt=1.)
    nmpc = NMPC(m)
    nmpc.quad_stage_cost.add_states(names=['P'], weights=[10.], ref=[2.])
    nmpc.horizon = 5
    with pytest.raises(ValueError, match="continuous time"):
        nmpc.setup(options={'integration_method': 'discrete'})
    with pytest.raises(NotImplementedError, match="degrees 1 to 4"):
        nmpc.setup(options={'degree': 5})
    nmpc.setup(options={'integration_method': 'rk4'})                  # explicit RK on the continuous model
    assert nmpc._n_v == 6 * 4 + 5 * 2


def test_collocation_full_batch_closed_loop():
    """BASELINE batch (1024) with the default transcription: everything converges, warm starts pay."""
    import torch
    spec = dict(C2)
    nmpc = _product(spec)
    x = torch.as_tensor(c2_x0(1024), device='cuda')
    p = torch.as_tensor(np.array(spec['p']), device='cuda')
    its = []
    for _ in range(4):
        u = nmpc.optimize(x, cp=p)
        its.append(float(nmpc._nlp_solution['iter_count'].double().mean()))
        assert np.all(nmpc.solver_status_code == 1), np.unique(nmpc.solver_status_code, return_counts=True)
        x = nmpc.plant_step(x, u, cp=p)
    assert its[-1] < its[0]


def test_explicit_runge_kutta_inside_the_nmpc_with_the_continuous_objective():
    """`integration_method='rk4'` on a CONTINUOUS model (the reference's own test, tests/test_NMPC.py `test_closed_loop_rk`):
    the Lagrange term is integrated with the scheme's weights at its stage points, quad = sum_i h b_i l(X_i, u)
    (modeling.py:1213-1281, the default objective of a continuous model, optimizer.py:1423-1426).  Checked without a second
    interior-point code: the reported objective equals that quadrature evaluated in numpy at the returned point, the defects
    vanish, and scipy SLSQP started at the returned point on the same NLP does not find a better one."""
    from scipy.optimize import minimize
    from hilo_mpc_amd import NMPC, Model
    from oracle import models
    N, h = 6, .1
    m = Model('pendulum4').setup(dt=h)
    nmpc = NMPC(m)
    nmpc.quad_stage_cost.add_states(names=['v', 'theta'], ref=[0, 0], weights=[10, 5])
    nmpc.quad_stage_cost.add_inputs(names='F', weights=0.1)
    nmpc.horizon = N
    nmpc.set_box_constraints(x_ub=[5, 10, 10, 10], x_lb=[-5, -10, -10, -10])
    nmpc.set_initial_guess(x_guess=[2.5, 0., .1, 0.], u_guess=0.)
    nmpc.setup(options={'integration_method': 'rk4'}, solver_options={'ipopt.tol': 1e-10})
    assert nmpc._nlp_options['objective_function'] == 'continuous' and nmpc._jit
    x0 = np.array([[2.5, 0., .1, 0.]])
    nmpc.optimize(x0)
    assert nmpc.solver_status_code[0] == 1
    v = nmpc._nlp_solution['x'].cpu().numpy()[0]
    om = models.get('pendulum4')
    f = lambda x, u: om.f(x[None], np.atleast_1d(u)[None], np.zeros((1, 0)), h)[0]           # noqa: E731
    A = [[0, 0, 0, 0], [.5, 0, 0, 0], [0, .5, 0, 0], [0, 0, 1., 0]]
    b = [1 / 6, 1 / 3, 1 / 3, 1 / 6]

    def lag(x, u):
        return 10 * x[1] ** 2 + 5 * x[2] ** 2 + .1 * u ** 2

    def rollout(w):
        X = np.concatenate([x0[0], w[:N * 4]]).reshape(N + 1, 4)
        U = w[N * 4:]
        J, c = 0., []
        for k in range(N):
            ks, q = [], 0.
            for i in range(4):
                Xi = X[k] + h * sum(A[i][j] * ks[j] for j in range(i))
                ks.append(f(Xi, U[k]))
                q += h * b[i] * lag(Xi, U[k])
            J += q
            c.append(X[k + 1] - (X[k] + h * sum(b[i] * ks[i] for i in range(4))))
        return J, np.concatenate(c)

    w = v[4:]
    J, c = rollout(w)
    np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy()[0], J, rtol=1e-10)
    assert np.abs(c).max() < 1e-9
    lb = np.concatenate([np.tile([-5, -10, -10, -10], N), np.full(N, -np.inf)])
    ub = np.concatenate([np.tile([5, 10, 10, 10], N), np.full(N, np.inf)])
    sol = minimize(lambda q: rollout(q)[0], w, method='SLSQP', bounds=list(zip(lb, ub)),
                   constraints=[{'type': 'eq', 'fun': lambda q: rollout(q)[1]}], options={'ftol': 1e-12, 'maxiter': 100})
    assert sol.fun >= J - 1e-7 * abs(J) and np.abs(sol.x - w).max() < 5e-4


def test_hard_terminal_constraint_under_collocation_vs_oracle():
    """A hard terminal constraint with the reference's default integration method (mpc.py:1693-1700 in the collocation branch: rows
    on the end state of the last interval, in g between its continuity rows and its node rows) - an EQUALITY (lb = ub) on a product
    of two states and a lower bound, both binding.  The engine imposes the rows on the integrated end state like the reference; the
    output pass adds their pull to the collocation rows of the last interval.  v 1e-6, objective 1e-9, multipliers 1e-5."""
    from oracle.nmpc import IpmOptions
    from oracle.nmpc_coll_gen import GenCollIpm
    from tests.problems import oracle_coll_gen, product_gen
    spec = dict(C2, N=6, collocation=dict(degree=3))
    x0 = c2_x0(4)
    free = GenCollIpm(oracle_coll_gen(spec), IpmOptions(tol=1e-10))
    xN = free.solve(x0, spec['p'])['X'][:, -1] * free.pb.sx
    prod, smin = float((xN[:, 3] * xN[:, 1]).mean()), float(xN[:, 1].max() * 1.002)
    spec['terminal_constraint'] = dict(expr=['I*S', 'S'], lb=[prod, smin], ub=[prod, np.inf])
    pb = oracle_coll_gen(spec)
    ipm = GenCollIpm(pb, IpmOptions(tol=1e-10))
    ref = ipm.solve(x0, spec['p'])
    assert np.all(ref['status'] == 1)
    nmpc = product_gen(spec, **{'ipopt.tol': 1e-10})
    assert (nmpc._n_v, nmpc._n_g) == (pb.n_v, pb.n_g)
    u = nmpc.optimize(x0, cp=spec['p'])
    assert np.all(nmpc.solver_status_code == 1)
    v, vr = nmpc._nlp_solution['x'].cpu().numpy(), ipm.to_v(ref)
    assert np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))) < 1e-6
    np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy(), ref['f'], rtol=1e-9)
    np.testing.assert_allclose(u, ref['u0'], rtol=1e-6, atol=1e-8)
    lam, lr = nmpc._nlp_solution['lam_g'].cpu().numpy(), ipm.lam_g(ref)
    assert np.max(np.abs(lam - lr) / np.maximum(1., np.abs(lr))) < 1e-5
    per = (pb.n_g - 2) // pb.N
    lt = lam[:, (pb.N - 1) * per + 3 * 4 + 4:(pb.N - 1) * per + 3 * 4 + 4 + 2]
    assert np.all(np.abs(lt[:, 0]) > 1e-4)                                  # the equality carries force
    xe = v[:, pb.x_ind[-1]]
    assert np.all(np.abs(xe[:, 3] * xe[:, 1] - prod) < 1e-6 * prod) and np.all(xe[:, 1] >= smin - 1e-6)


def _racecar(N, **solver_options):
    """The reference's minimum-time test (tests/test_NMPC.py:2706-2760) with a shorter horizon."""
    from hilo_mpc_amd import NMPC, Model
    from hilo_mpc_amd.expr import sin
    m = Model(name='racecar')
    x = m.set_dynamical_states(['p', 'v'])
    u = m.set_inputs(['u'])
    m.set_dynamical_equations([x[1], u[0] - x[1]])
    m.setup(dt=.1)
    nmpc = NMPC(m)
    nmpc.minimize_final_time(weight=1)
    nmpc.horizon = N
    nmpc.set_initial_guess(x_guess=[0, 0], u_guess=0.)
    nmpc.set_terminal_constraints(terminal_constraint=m.x[0], lb=1, ub=1)
    nmpc.set_stage_constraints(stage_constraint=m.x[1] - (1 - sin(2 * np.pi * m.x[0]) / 2), lb=-np.inf, ub=0)
    nmpc.set_box_constraints(u_lb=0, u_ub=1)
    nmpc.setup(solver_options=solver_options or None)
    return nmpc


def test_minimum_time_race_car_vs_oracle():
    """`minimize_final_time` (mpc.py:859-866, :1606-1617, :1746-1754; the reference's own test tests/test_NMPC.py:2706-2760): the N
    sampling intervals as variables forced equal, J = sum(dt), a terminal EQUALITY p(N) = 1, the speed limit v <= 1 - sin(2 pi p) / 2
    at every collocation point and node, default options (collocation).  The product carries the common interval as a state r with
    dt = r^2 (nmpc.py::_setup_min_time) and hands out the reference's layout: v with the dt block last, lam_g with the N - 1 dt rows
    last.  Same minimiser as the oracle's NLP with dt variables (oracle/nmpc_coll_gen.py, min_time=): v 1e-6, f 1e-8, multipliers 1e-5;
    the final time is the known ~1.9 s of this example."""
    from oracle.nmpc import IpmOptions
    from oracle.nmpc_coll_gen import GenCollIpm, GenCollProblem
    N = 20
    pb = GenCollProblem(models.get('racecar2'), dt=.1, N=N, degree=3, constraint=dict(expr=['v - (1 - sin(2*pi*p)/2)'], lb=[-np.inf], ub=[0.]),
                        terminal=dict(expr=['p'], lb=[1.], ub=[1.]), min_time=1., u_lb=[0.], u_ub=[1.], x_guess=[0., 0.], u_guess=[0.])
    ipm = GenCollIpm(pb, IpmOptions(tol=1e-10))
    x0 = np.array([[0., 0.], [.02, .1]])
    ref = ipm.solve(x0, [])
    assert np.all(ref['status'] == 1) and abs(ref['dt'][0].sum() - 1.9065) < 1e-3
    nmpc = _racecar(N, **{'ipopt.tol': 1e-10})
    assert (nmpc._n_v, nmpc._n_g) == (pb.n_v, pb.n_g) and nmpc._dt_ind == pb.dt_ind and nmpc._x_ind == pb.x_ind and nmpc._u_ind == pb.u_ind
    u = nmpc.optimize(x0)
    assert np.all(nmpc.solver_status_code == 1)
    v, vr = nmpc._nlp_solution['x'].cpu().numpy(), ipm.to_v(ref)
    assert np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))) < 1e-6
    np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy(), ref['f'], rtol=1e-8)
    np.testing.assert_allclose(u, ref['u0'], rtol=1e-6, atol=1e-7)
    lam, lr = nmpc._nlp_solution['lam_g'].cpu().numpy(), ipm.lam_g(ref)
    assert np.max(np.abs(lam - lr) / np.maximum(1., np.abs(lr))) < 1e-5
    xp, up, dtp = nmpc.return_prediction()
    assert dtp.shape == (2, N) and np.allclose(dtp, dtp[:, :1]) and abs(dtp[0].sum() - ref['dt'][0].sum()) < 1e-6
    assert abs(xp[0, 0, -1] - 1.) < 1e-7                                   # the car is at p = 1 at the final time


def test_minimum_time_with_the_reference_test_horizon():
    """N = 100 as in tests/test_NMPC.py:2745 (no oracle at this size): solved, intervals equal, terminal position reached, the speed
    limit kept at the nodes, final time within 1 % of the N = 20 problem's."""
    nmpc = _racecar(100)
    nmpc.optimize([0., 0.])
    assert nmpc.solver_status_code[0] == 1
    xp, up, dtp = nmpc.return_prediction()
    assert np.allclose(dtp, dtp[:, :1], rtol=1e-9) and abs(dtp.sum() - 1.9065) < .02 and abs(xp[0, 0, -1] - 1.) < 1e-6
    assert np.all(xp[0, 1] - (1 - np.sin(2 * np.pi * xp[0, 0]) / 2) < 1e-6) and np.all((up >= -1e-8) & (up <= 1 + 1e-8))
