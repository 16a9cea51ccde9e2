"""GPU tests of host-side features added after the last GPU session of round 2 (their device paths are the verified ones of
the general policy and of the particle-filter function): kept in the LAST test module so that the established suite runs first."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.problems import C2, c2_x0          # noqa: E402


def test_measurement_box_constraints_equal_explicit_stage_and_terminal_constraints():
    """`set_box_constraints(y_ub=)` (mpc.py:703-708) IS a stage + terminal constraint on the measurement equations: identical
    solves, and the bound holds along the prediction where the solver converged."""
    from hilo_mpc_amd import NMPC, Model

    def build(as_y):
        m = Model('chemostat4').discretize('rk4').setup(dt=C2['dt'])
        nmpc = NMPC(m)
        xs, us = m.dynamical_state_names, m.input_names
        for ind, W, ref in C2['stage_states']:
            nmpc.quad_stage_cost.add_states(names=[xs[i] for i in ind], weights=list(W), ref=ref)
        for ind, W, ref in C2['stage_inputs']:
            nmpc.quad_stage_cost.add_inputs(names=[us[i] for i in ind], weights=list(W), ref=ref)
        for ind, W, ref in C2['terminal_states']:
            nmpc.quad_terminal_cost.add_states(names=[xs[i] for i in ind], weights=list(W), ref=ref)
        nmpc.horizon = 8
        box = dict(x_lb=C2.get('x_lb'), u_lb=C2.get('u_lb'), u_ub=C2.get('u_ub'))
        if as_y:
            nmpc.set_box_constraints(y_ub=[10., .45], **box)                    # yX = X, yP = P
        else:
            nmpc.set_box_constraints(**box)
            nmpc.set_stage_constraints(stage_constraint=[m.x['X'], m.x['P']], ub=[10., .45])
            nmpc.set_terminal_constraints(terminal_constraint=[m.x['X'], m.x['P']], ub=[10., .45])
        nmpc.set_initial_guess(x_guess=C2['x_guess'], u_guess=C2['u_guess'])
        nmpc.setup(options={'integration_method': 'discrete', 'print_level': 0})
        return nmpc

    a, b = build(True), build(False)
    assert (a._n_v, a._n_g) == (b._n_v, b._n_g) and a._user_source == b._user_source
    x0 = c2_x0(6)
    ua, ub = a.optimize(x0, cp=C2['p']), b.optimize(x0, cp=C2['p'])
    assert np.array_equal(a.solver_status_code, b.solver_status_code)
    np.testing.assert_array_equal(ua, ub)
    np.testing.assert_array_equal(a._nlp_solution['x'].cpu().numpy(), b._nlp_solution['x'].cpu().numpy())
    ok = a.solver_status_code == 1
    xp = a.return_prediction()[0]
    assert np.all(xp[ok][:, 2, :-1] <= .45 + 1e-6)


@pytest.mark.parametrize('name', ['toy1d', 'chemostat4', 'pend1'])
def test_model_step_vs_oracle(name):
    """`Model.step` (the plant side of the closed loop: hilo_pf_function with one particle per instance and no noise) against the
    oracle's discretised model and measurement map; `simulate` / `solution` on top of it."""
    from tests.test_pf_gpu import POINT, _case
    m, om, dt = _case(name)
    x0, u, p = POINT[name]
    rng = np.random.default_rng(4)
    B = 37
    X = np.asarray(x0) * (1 + .1 * rng.standard_normal((B, m.n_x)))
    U = np.tile(np.asarray(u, dtype=float), (B, 1)) * (1 + .1 * rng.standard_normal((B, len(u)))) if len(u) else None
    xn, y = m.step(X, U, p if len(p) else None)
    Ur = U if U is not None else np.zeros((B, 0))
    Pr = np.tile(np.asarray(p, dtype=float), (B, 1)) if len(p) else np.zeros((B, 0))
    xr = om.f(X, Ur, Pr, dt)
    np.testing.assert_allclose(xn, xr, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(y, om.h(xr, Ur, Pr, dt), rtol=1e-11, atol=1e-13)
    m.set_initial_conditions(X[0])
    m.simulate(u=None if U is None else U[0], p=p if len(p) else None, steps=2)
    x2 = om.f(xr[:1], Ur[:1], Pr[:1], dt)
    np.testing.assert_allclose(m.solution['x:f'][:, 0], x2[0], rtol=1e-10, atol=1e-12)
    assert m.solution['x'].shape == (m.n_x, 3)


def test_closed_loop_with_a_plant_model_of_its_own():
    """`SimpleControlLoop(plant, controller)` with a plant that is NOT the controller's model (control_loop.py:343-397): here the
    same equations advanced with the controller's discretisation - the loop must reproduce the controller's own plant step."""
    from hilo_mpc_amd import Model, SimpleControlLoop
    from tests.problems import product_nmpc
    nmpc = product_nmpc(C2)
    plant = Model('chemostat4').discretize('rk4').setup(dt=C2['dt'])
    x0 = c2_x0(4)
    a = SimpleControlLoop(plant, nmpc).run(3, x0, p=C2['p'])
    b = SimpleControlLoop(nmpc._model, product_nmpc(C2)).run(3, x0, p=C2['p'])
    np.testing.assert_allclose(a['x'], b['x'], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(a['u'], b['u'], rtol=1e-8, atol=1e-10)


def test_model_given_as_text_equals_the_zoo_model():
    """`Model.set_equations(equations=<text>)` (the benchmark system of tests/test_PFs.py:39-44, with `dt` in the difference
    equation): its compiled functor steps and measures like the zoo's Toy1D."""
    from hilo_mpc_amd import Model
    m = Model(name='toy_text', discrete=True)
    m.set_equations(equations=['x(k+1) = x(k)/2 + 25*dt*x(k)/(1 + x(k)^2)', 'y(k) = x(k)^2/20'])
    m.setup(dt=1.)
    zoo = Model('toy1d').setup(dt=1.)
    X = np.linspace(-6., 6., 41)[:, None]
    xa, ya = m.step(X)
    xb, yb = zoo.step(X)
    np.testing.assert_allclose(xa, xb, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(ya, yb, rtol=1e-13, atol=1e-15)
