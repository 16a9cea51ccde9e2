"""CPU: the NMPC oracle is 'parity unpinned' (no reference numbers exist); it is cross-checked here by an independent
solver (scipy SLSQP) on the same NLP in the reference's own layout (terminal cost on Phi_{N-1}, x_0 pinned by its
bounds) and by its KKT residual.  Also pins the integer bookkeeping of the transcription."""
import numpy as np
import pytest
from scipy.optimize import minimize

from oracle.nmpc import DenseIpm, NmpcProblem
from oracle import models
from tests.problems import C2, c2_x0, oracle_problem


def test_transcription_layout_c2():
    pb = oracle_problem(C2)
    # n_v = Nc nu + (N+1) nx = 40 + 84, n_g = N nx
    assert (pb.n_v, pb.n_g) == (124, 80)
    assert pb.x_ind[0] == [0, 1, 2, 3] and pb.x_ind[20] == [80, 81, 82, 83]
    assert pb.u_ind[0] == [84, 85] and pb.u_ind[19] == [122, 123]
    assert pb.v_lb[84] == 0. and pb.v_ub[85] == 1. and np.isinf(pb.v_ub[0])
    np.testing.assert_array_equal(pb.v_guess[:4], [.1, 40., 0., 0.])


@pytest.mark.parametrize('spec_over', [dict(N=6), dict(N=6, x_scaling=[1., 10., 1., 1.], u_scaling=[.5, 2.])])
def test_dense_ipm_vs_slsqp(spec_over):
    spec = dict(C2)
    spec.update(spec_over)
    pb = oracle_problem(spec)
    ipm = DenseIpm(pb)
    x0 = c2_x0(2)
    res = ipm.solve(x0, spec['p'])
    assert np.all(res['status'] == 1) and np.all(res['kkt'] <= 1e-8)
    nx = pb.nx
    for b in range(2):
        # x_0 is pinned by lb = ub (mpc.py:801-802); scipy's finite differences cannot step a fixed variable, so the
        # independent solver works on the remaining variables with x_0 substituted - the same NLP
        x0s = x0[b] / pb.sx
        full = lambda w: np.concatenate([x0s, w])                                   # noqa: E731
        lb, ub = pb.v_lb[nx:], pb.v_ub[nx:]
        v_ipm = ipm.to_v(res)[b]
        w_start = np.clip(v_ipm[nx:] + 1e-3 * np.random.default_rng(b).normal(size=pb.n_v - nx), lb, ub)
        sol = minimize(lambda w: pb.objective(full(w), spec['p'])[0], w_start, method='SLSQP',
                       bounds=list(zip(lb, ub)),
                       constraints=[{'type': 'eq', 'fun': lambda w: pb.constraints(full(w), spec['p'])[0]}],
                       options={'ftol': 1e-10, 'maxiter': 500})
        assert sol.success, sol.message
        # SLSQP terminates on ftol with finite-difference gradients: it pins the solution to ~1e-4, the objective
        # to ~1e-7
        np.testing.assert_allclose(sol.fun, pb.objective(v_ipm, spec['p'])[0], rtol=1e-6)
        np.testing.assert_allclose(sol.x, v_ipm[nx:], rtol=2e-4, atol=2e-4)
        assert np.abs(pb.constraints(v_ipm, spec['p'])).max() < 1e-8


def test_input_change_term_and_pendulum():
    """Non-convex dynamics + input-change penalty (only interval 0, mpc.py:1631-1635)."""
    pb = NmpcProblem(models.get('pendulum4'), dt=.1, N=10, order=4,
                     stage_states=[([1, 2], [10., 5.], [0., 0.])], stage_inputs=[([0], [.1], None)],
                     input_change=([0], [1.]),
                     x_lb=[-5, -10, -10, -10], x_ub=[5, 10, 10, 10], x_guess=[2.5, 0., .1, 0.], u_guess=[0.])
    ipm = DenseIpm(pb)
    res = ipm.solve(np.array([[2.5, 0., .1, 0.]]), np.zeros((1, 0)), u_old=np.array([[.3]]))
    assert res['status'][0] == 1 and res['kkt'][0] <= 1e-8
    v = ipm.to_v(res)[0]
    f_with = pb.objective(v, np.zeros((1, 0)), u_old=np.array([[.3]]))[0]
    f_without = pb.objective(v, np.zeros((1, 0)))[0]
    np.testing.assert_allclose(f_with - f_without, (v[pb.u_ind[0][0]] - .3) ** 2, rtol=1e-10)


def test_control_horizon_shorter_than_prediction_horizon_vs_slsqp():
    """mpc.py:1476-1485, :1629-1630 (Nc input blocks, the last held): the interior-point solution of the Nc < N problem
    against scipy SLSQP on the reference-layout NLP functions."""
    from scipy.optimize import minimize
    from tests.problems import C2, c2_x0, oracle_problem
    spec = dict(C2, N=8, Nc=3)
    pb = oracle_problem(spec)
    assert pb.n_v == 9 * 4 + 3 * 2 and len(pb.u_ind) == 3
    ipm = DenseIpm(pb)
    x0 = c2_x0(2)
    r = ipm.solve(x0, spec['p'])
    assert np.all(r['status'] == 1)
    v = ipm.to_v(r)
    p = np.array(spec['p'])
    lb, ub = pb.v_lb.copy(), pb.v_ub.copy()
    lb[:4] = ub[:4] = x0[0] / pb.sx
    res = minimize(lambda q: pb.objective(q[None], p[None])[0], v[0] + 1e-3,
                   constraints={'type': 'eq', 'fun': lambda q: pb.constraints(q[None], p[None])[0]},
                   bounds=list(zip(lb, ub)), method='SLSQP', options={'ftol': 1e-14, 'maxiter': 500})
    # SLSQP may stop with 'positive directional derivative' at the optimum: judge it by what it found
    assert abs(res.fun - r['f'][0]) < 1e-6 * abs(r['f'][0]) and np.abs(res.x - v[0]).max() < 1e-4
    assert np.abs(pb.constraints(res.x[None], p[None])).max() < 1e-8
