"""GPU parity of NMPC on semi-explicit DAE models (row a1: algebraic states, mpc.py:1488-1527; the reference's own DAE test is
tests/test_NMPC.py:1866-1987): the product eliminates the algebraic states through their equations inside the run-time compiled
model and rebuilds them - and the multipliers of their rows - in the reference's layout; the oracle (oracle/nmpc_dae.py)
carries them as variables like the reference."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import models                                              # noqa: E402
from oracle.nmpc import IpmOptions                                     # noqa: E402
from oracle.nmpc_dae import DaeCollIpm, DaeCollProblem                 # noqa: E402
from tests.problems import C2, c2_x0, symbolic_model                  # noqa: E402


def _pendulum(tol=None):
    from hilo_mpc_amd import NMPC
    m = symbolic_model('pendulum4_dae').setup(dt=.1)
    nmpc = NMPC(m)
    nmpc.quad_stage_cost.add_states(names=['v', 'theta'], ref=[0, 0], weights=[10, 5])        # tests/test_NMPC.py:1925-1926
    nmpc.quad_stage_cost.add_inputs(names='F', weights=0.1)
    nmpc.horizon = 25
    nmpc.set_box_constraints(x_ub=[5, 10, 10, 10], x_lb=[-5, -10, -10, -10])
    nmpc.set_initial_guess(x_guess=[2.5, 0., .1, 0.], u_guess=0., z_guess=1.4)
    nmpc.setup(solver_options={'ipopt.tol': tol} if tol else None)                             # default: collocation, Radau 3
    return nmpc


def _pendulum_oracle(**opt):
    pb = DaeCollProblem(models.get('pendulum4_dae'), dt=.1, N=25, z_guess=[1.4], stage_states=[([1, 2], [10., 5.], [0., 0.])],
                        stage_inputs=[([0], [.1], None)], x_lb=[-5, -10, -10, -10], x_ub=[5, 10, 10, 10],
                        x_guess=[2.5, 0., .1, 0.], u_guess=[0.])
    return pb, DaeCollIpm(pb, IpmOptions(**opt))


def test_reference_dae_test_problem_layout_and_solution():
    """The pendulum DAE of the reference's test: index maps equal the reference's layout, the solution (incl. the algebraic
    states at the collocation points and the multipliers in the reference's row order) equals the oracle's."""
    x0 = np.array([[2.5, 0., .1, 0.], [2., .2, -.1, .1]])
    pb, ipm = _pendulum_oracle(tol=1e-10)
    nmpc = _pendulum(tol=1e-10)
    assert (nmpc._n_v, nmpc._n_g) == (pb.n_v, pb.n_g)
    assert nmpc._z_ind == pb.z_ind and nmpc._ip_ind == pb.ip_ind and nmpc._zp_ind == pb.zp_ind
    ref = ipm.solve(x0, [])
    u = nmpc.optimize(x0)
    assert np.array_equal(nmpc.solver_status_code, ref['status']) and np.all(ref['status'] == 1)
    v, vr = nmpc._nlp_solution['x'].cpu().numpy(), ipm.to_v(ref)
    assert np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))) < 1e-6
    np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy(), ref['f'], rtol=1e-10)
    np.testing.assert_allclose(u, ref['u0'], rtol=1e-6, atol=1e-8)
    lam = nmpc._nlp_solution['lam_g'].cpu().numpy()
    np.testing.assert_allclose(lam, ref['lam'], rtol=1e-5, atol=1e-7)
    # closed loop like the reference's test (plant = the model itself)
    x = x0
    for _ in range(3):
        u = nmpc.optimize(x)
        x = nmpc.plant_step(x, u).cpu().numpy()
    assert np.all(nmpc.solver_status_code == 1) and np.all(np.isfinite(x))


def test_algebraic_state_feeding_back_into_the_dynamics():
    """chemostat4 with the growth rate as algebraic state (oracle/models.py::chemostat4_dae) - the algebraic rows carry force;
    B = 64 instances."""
    from hilo_mpc_amd import NMPC
    spec = dict(C2, N=8)
    m = symbolic_model('chemostat4_dae').setup(dt=spec['dt'])
    nmpc = NMPC(m)
    xs, us = m.dynamical_state_names, m.input_names
    for ind, W, ref in spec['stage_states']:
        nmpc.quad_stage_cost.add_states(names=[xs[i] for i in ind], weights=list(W), ref=ref)
    for ind, W, ref in spec['stage_inputs']:
        nmpc.quad_stage_cost.add_inputs(names=[us[i] for i in ind], weights=list(W), ref=ref)
    for ind, W, ref in spec.get('terminal_states', []):
        nmpc.quad_terminal_cost.add_states(names=[xs[i] for i in ind], weights=list(W), ref=ref)
    nmpc.horizon = spec['N']
    nmpc.set_box_constraints(x_ub=spec.get('x_ub'), x_lb=spec.get('x_lb'), u_ub=spec.get('u_ub'), u_lb=spec.get('u_lb'))
    nmpc.set_initial_guess(x_guess=spec.get('x_guess'), u_guess=spec.get('u_guess'), z_guess=[.3])
    nmpc.setup(solver_options={'ipopt.tol': 1e-10})
    kw = {k: v for k, v in spec.items() if k not in ('model', 'p', 'order')}
    pb = DaeCollProblem(models.get('chemostat4_dae'), z_guess=[.3], **kw)
    ipm = DaeCollIpm(pb, IpmOptions(tol=1e-10))
    x0 = c2_x0(64)
    ref = ipm.solve(x0[:4], spec['p'])
    u = nmpc.optimize(x0, cp=spec['p'])
    assert np.all(nmpc.solver_status_code == 1) and np.all(ref['status'] == 1)
    v, vr = nmpc._nlp_solution['x'].cpu().numpy()[:4], ipm.to_v(ref)
    assert np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))) < 1e-6
    lam = ref['lam'].reshape(4, pb.N, -1).copy()
    lam[:, -1, -pb.nx:] += 2 * (ref['X'][:, -1] - pb.xrefN) @ pb.WN                # terminal cost on the end state (mpc.py:1682)
    np.testing.assert_allclose(nmpc._nlp_solution['lam_g'].cpu().numpy()[:4].reshape(4, pb.N, -1), lam, rtol=1e-5, atol=1e-7)
    assert np.abs(lam[:, :, [4, 9, 14]]).max() > 1e-3                               # the algebraic rows carry force here
    np.testing.assert_allclose(u[:4], ref['u0'], rtol=1e-6, atol=1e-8)


def test_unsupported_dae_configurations_are_refused():
    from hilo_mpc_amd import NMPC
    m = symbolic_model('pendulum4_dae').setup(dt=.1)
    nmpc = NMPC(m)
    nmpc.quad_stage_cost.add_states(names=['v'], ref=[0], weights=[1])
    nmpc.horizon = 5
    md = symbolic_model('pendulum4_dae').discretize('rk4').setup(dt=.1)
    n2 = NMPC(md)
    n2.quad_stage_cost.add_states(names=['v'], ref=[0], weights=[1])
    n2.horizon = 5
    with pytest.raises(NotImplementedError, match="collocation"):
        n2.setup(options={'integration_method': 'discrete'})


def test_bounds_on_algebraic_states_vs_oracle():
    """`set_box_constraints(z_lb=, z_ub=)` (mpc.py:645-701: the box of the zp blocks of v; the reference's own DAE test sets one,
    tests/test_NMPC.py:1964): the product eliminates the algebraic states, so the box becomes hard rows on z(x_{k,i}, u_k) at the
    collocation points; the oracle bounds its variables.  The pendulum's tip height y >= 1.494 - without it the pendulum is let
    fall towards the end of the horizon, with it the bound is active there; n_v / n_g are the unbounded problem's."""
    from hilo_mpc_amd import NMPC
    from oracle.nmpc_coll_gen import GenCollIpm, GenCollProblem
    m = symbolic_model('pendulum4_dae').setup(dt=.1)
    nmpc = NMPC(m)
    nmpc.quad_stage_cost.add_states(names=['v', 'theta'], ref=[0, 0], weights=[10, 5])
    nmpc.quad_stage_cost.add_inputs(names='F', weights=0.1)
    nmpc.horizon = 12
    nmpc.set_box_constraints(x_ub=[5, 10, 10, 10], x_lb=[-5, -10, -10, -10], z_lb=1.494)
    nmpc.set_initial_guess(x_guess=[2.5, 0., .1, 0.], u_guess=0., z_guess=1.4)
    nmpc.setup(solver_options={'ipopt.tol': 1e-10})
    kw = dict(dt=.1, N=12, z_guess=[1.4], stage_states=[([1, 2], [10., 5.], [0., 0.])], stage_inputs=[([0], [.1], None)],
              x_lb=[-5, -10, -10, -10], x_ub=[5, 10, 10, 10], x_guess=[2.5, 0., .1, 0.], u_guess=[0.])
    pb = GenCollProblem(models.get('pendulum4_dae'), z_lb=[1.494], **kw)
    ipm = GenCollIpm(pb, IpmOptions(tol=1e-10))
    free = GenCollIpm(GenCollProblem(models.get('pendulum4_dae'), **kw), IpmOptions(tol=1e-10))
    x0 = np.array([[2.5, 0., .1, 0.], [2., .2, -.1, .1]])
    ref = ipm.solve(x0, [])
    assert np.all(ref['status'] == 1) and ref['Zc'].min() < 1.494 + 1e-6 and free.solve(x0, [])['Zc'].min() < 1.1     # the bound is active
    assert (nmpc._n_v, nmpc._n_g) == (pb.n_v, pb.n_g)
    u = nmpc.optimize(x0)
    assert np.all(nmpc.solver_status_code == 1)
    v, vr = nmpc._nlp_solution['x'].cpu().numpy(), ipm.to_v(ref)
    assert np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))) < 1e-6
    np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy(), ref['f'], rtol=1e-9)
    np.testing.assert_allclose(u, ref['u0'], rtol=1e-6, atol=1e-8)
    zp = v[:, [i for k in range(pb.N) for i in nmpc._zp_ind[k]]]
    assert zp.min() >= 1.494 - 1e-7 and zp.min() < 1.494 + 1e-6
