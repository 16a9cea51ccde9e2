"""GPU: `NMPC.set_custom_constraints_function` (optimizer.py:1180-1208; rows `lb <= fun(v, x_ind, u_ind) <= ub` at the END of g,
mpc.py:1729-1745) - a function of the whole decision vector, offloaded for sums over the stages of single-stage terms
(hilo_mpc_amd/custom.py): each row rides on an accumulator state of the stage-structured problem.  The oracle (oracle/nmpc_gen.py)
keeps the rows as what they are in the reference: dense rows over v - so the comparison is not self-referential."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.nmpc import IpmOptions                                             # noqa: E402
from oracle.nmpc_gen import GenIpm                                              # noqa: E402
from tests.problems import C2, c2_x0, oracle_gen, product_gen                 # noqa: E402


def _trapezoid(i, dt):
    def fun(v, x_ind, u_ind):
        s = 0
        for k in range(len(x_ind) - 1):
            s += (v[x_ind[k][i]] + v[x_ind[k + 1][i]]) / 2 * dt            # tests/test_NMPC.py:524-529
        return s
    return fun


def test_trapezoid_integral_bound_vs_dense_oracle():
    """Tracking NMPC on the chemostat, N = 8: the trapezoid integral of the product concentration over the horizon bounded at 90 %
    of its unconstrained value - status, v, f, u_0 and lam_g incl. the custom row's multiplier (last entry) and the multipliers of
    the LAST shooting defect, which in the reference carry - nu dc/dx_N (the row acts on the node variable x_N there)."""
    spec = dict(C2, N=8)
    x0 = c2_x0(6)
    free = GenIpm(oracle_gen(spec)).solve(x0, C2['p'])
    X = free['X']
    ub = float(((X[:, :-1, 2] + X[:, 1:, 2]) / 2 * spec['dt']).sum(1).min() * .9)
    spec = dict(spec, custom=dict(fun=_trapezoid(2, spec['dt']), lb=0., ub=ub))
    pb = oracle_gen(spec)
    ipm = GenIpm(pb, IpmOptions(tol=1e-10))
    ref = ipm.solve(x0, C2['p'])
    assert np.all(ref['status'] == 1)
    nmpc = product_gen(spec, tol=1e-10)
    assert nmpc._jit and nmpc._nq == 1 and (nmpc._n_v, nmpc._n_g) == (pb.n_v, pb.n_g)
    nmpc.keep_full_solution = True
    u = nmpc.optimize(x0, cp=C2['p'])
    assert np.array_equal(nmpc.solver_status_code, ref['status'])
    v, vr = nmpc._nlp_solution['x'].cpu().numpy(), ipm.to_v(ref)
    assert v.shape == vr.shape and np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))) < 1e-6
    np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy(), ref['f'], rtol=1e-9)
    np.testing.assert_allclose(u, ref['u0'], rtol=1e-6, atol=1e-8)
    Xp = v[:, :(spec['N'] + 1) * 4].reshape(-1, spec['N'] + 1, 4)
    integral = ((Xp[:, :-1, 2] + Xp[:, 1:, 2]) / 2 * spec['dt']).sum(1)
    np.testing.assert_allclose(integral, ub, rtol=1e-7)                      # active in every instance
    lam, lr = nmpc._nlp_solution['lam_g'].cpu().numpy(), ipm.lam_g(ref)
    lr = lr.copy()
    last = slice((spec['N'] - 1) * 4, spec['N'] * 4)
    lr[:, last] += 2 * (ref['X'][:, -1] - pb.xrefNa) @ pb.WNa               # terminal cost convention (mpc.py:1682), as in test_gen_gpu
    assert lam.shape == lr.shape
    np.testing.assert_allclose(lam[:, -1], lr[:, -1], rtol=1e-5)              # the custom row
    assert np.all(lam[:, -1] > 1.)
    np.testing.assert_allclose(lam, lr, rtol=2e-5, atol=1e-5 * np.abs(lr).max())
    g = nmpc._nlp_solution['g'].cpu().numpy()
    np.testing.assert_allclose(g[:, -1], integral, rtol=1e-9)                # value of the row in g's last position


def test_two_rows_with_nonlinear_stage_terms_and_a_constant():
    """Two rows: the trapezoid integral (active) and an input-energy budget with a product of a state and an input and a constant
    part - three stage expressions shared over the stages; against the dense oracle."""
    spec = dict(C2, N=6)
    x0 = c2_x0(4)

    def fun(v, x_ind, u_ind):
        e = 0.5
        for k in range(len(u_ind)):
            e = e + 2. * v[u_ind[k][0]] ** 2 + v[x_ind[k][0]] * v[u_ind[k][1]] / 10
        return [_trapezoid(2, spec['dt'])(v, x_ind, u_ind), e]
    free = GenIpm(oracle_gen(spec)).solve(x0, C2['p'])
    X, U = free['X'], free['U']
    ub0 = float(((X[:, :-1, 2] + X[:, 1:, 2]) / 2 * spec['dt']).sum(1).min() * .92)
    e_free = .5 + (2 * U[:, :, 0] ** 2 + X[:, :-1, 0] * U[:, :, 1] / 10).sum(1)
    spec = dict(spec, custom=dict(fun=fun, lb=[0., -np.inf], ub=[ub0, float(.5 + .6 * (e_free.min() - .5))]))      # both rows active
    pb = oracle_gen(spec)
    ipm = GenIpm(pb, IpmOptions(tol=1e-10))
    ref = ipm.solve(x0, C2['p'])
    assert np.all(ref['status'] == 1)
    nmpc = product_gen(spec, tol=1e-10)
    assert nmpc._nq == 2 and (nmpc._n_v, nmpc._n_g) == (pb.n_v, pb.n_g)
    u = nmpc.optimize(x0, cp=C2['p'])
    assert np.array_equal(nmpc.solver_status_code, ref['status'])
    v, vr = nmpc._nlp_solution['x'].cpu().numpy(), ipm.to_v(ref)
    assert np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))) < 1e-6
    np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy(), ref['f'], rtol=1e-9)
    np.testing.assert_allclose(u, ref['u0'], rtol=1e-6, atol=1e-8)
    lam, lr = nmpc._nlp_solution['lam_g'].cpu().numpy(), ipm.lam_g(ref)
    np.testing.assert_allclose(lam[:, -2:], lr[:, -2:], rtol=1e-5, atol=1e-7)


def test_soft_custom_rows_vs_dense_oracle():
    """`soft=True` (mpc.py:1551-1556, :1731-1740): one slack e_cus per row behind the other slacks in v, 1e4 e_cus^T e_cus in the
    objective, two rows per function at the end of g (fun - e_cus <= ub, then fun + e_cus >= lb).  Two rows: the trapezoid integral
    with a tight upper bound (its slack opens) and the sum of the second input over the horizon bounded from BELOW only (linear: one
    minimum; its `- e` row is not imposed and keeps a zero multiplier); against the dense oracle - v incl. both slacks, f, u_0, lam_g."""
    spec = dict(C2, N=6)
    x0 = c2_x0(4)

    def fun(v, x_ind, u_ind):
        e = 0
        for k in range(len(u_ind)):
            e = e + v[u_ind[k][1]]
        return [_trapezoid(2, spec['dt'])(v, x_ind, u_ind), e]
    free = GenIpm(oracle_gen(spec)).solve(x0, C2['p'])
    X, U = free['X'], free['U']
    ub0 = float(((X[:, :-1, 2] + X[:, 1:, 2]) / 2 * spec['dt']).sum(1).min() * .9)
    lb1 = float(U[:, :, 1].sum(1).max() * 2.45)          # (the first row alone moves the sum from 0.073 to 0.11 - 0.13)
    spec = dict(spec, custom=dict(fun=fun, lb=[0., lb1], ub=[ub0, np.inf], soft=True, max_violation=[5., 7.]))
    pb = oracle_gen(spec)
    ipm = GenIpm(pb, IpmOptions(tol=1e-10))
    ref = ipm.solve(x0, C2['p'])
    assert np.all(ref['status'] == 1)
    nmpc = product_gen(spec, tol=1e-10)
    assert nmpc._jit and nmpc._nq == 2 and (nmpc._n_v, nmpc._n_g) == (pb.n_v, pb.n_g)
    assert nmpc._e_cus_ind == list(range(pb.n_v - 2, pb.n_v))
    nmpc.keep_full_solution = True
    u = nmpc.optimize(x0, cp=C2['p'])
    assert np.array_equal(nmpc.solver_status_code, ref['status'])
    v, vr = nmpc._nlp_solution['x'].cpu().numpy(), ipm.to_v(ref)
    assert v.shape == vr.shape and np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))) < 1e-6
    assert np.all(v[:, -2] > 1e-4) and np.all(v[:, -1] > 1e-5)                # both slacks open
    np.testing.assert_allclose(v[:, -2:], vr[:, -2:], rtol=1e-5)
    np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy(), ref['f'], rtol=1e-9)
    np.testing.assert_allclose(u, ref['u0'], rtol=1e-6, atol=1e-8)
    lam, lr = nmpc._nlp_solution['lam_g'].cpu().numpy(), ipm.lam_g(ref)
    assert lam.shape == lr.shape
    # the four custom rows: [fun_0 - e_0 <= ub (active), fun_1 - e_1 <= inf (not imposed), fun_0 + e_0 >= 0 (inactive), fun_1 + e_1 >= lb (active)]
    np.testing.assert_allclose(lam[:, -4:], lr[:, -4:], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(lam[:, -4], 2e4 * v[:, -2], rtol=1e-5)         # stationarity in e_0: 2 W e = multiplier
    np.testing.assert_allclose(lam[:, -1], -2e4 * v[:, -1], rtol=1e-5)
    assert np.all(lam[:, -3] == 0.)
    g = nmpc._nlp_solution['g'].cpu().numpy()
    Xp = v[:, :(spec['N'] + 1) * 4].reshape(-1, spec['N'] + 1, 4)
    integral = ((Xp[:, :-1, 2] + Xp[:, 1:, 2]) / 2 * spec['dt']).sum(1)
    np.testing.assert_allclose(g[:, -4], integral - v[:, -2], rtol=1e-9)      # fun_0 - e_0
    np.testing.assert_allclose(g[:, -2], integral + v[:, -2], rtol=1e-9)      # fun_0 + e_0
    np.testing.assert_allclose(integral - v[:, -2], ub0, rtol=1e-7)


def test_the_reference_test_case_under_the_default_collocation():
    """tests/test_NMPC.py:519-552 as written: the cart pendulum with the `dummy` integrator state (x_5' = u_dummy), tracking
    theta -> pi and dummy -> 10 over N = 10 under the default transcription (collocation, Radau 3, continuous objective), the
    trapezoid integral of `dummy` over the nodes in [0, 4].  The reference asserts `integral - 4 < 1e-3`; here also: solved, the
    row active (the unconstrained problem drives dummy towards 10: integral 4.6), its multiplier positive."""
    from hilo_mpc_amd import NMPC, Model
    from hilo_mpc_amd.expr import cos, sin
    M, m_, l_, g_ = 5., 1., 1., 9.81
    model = Model()
    x = model.set_dynamical_states(['x', 'v', 'theta', 'omega', 'dummy'])
    F = model.set_inputs(['F', 'u_dummy'])
    v, theta, omega = x[1], x[2], x[3]
    dv = 1. / (M + m_ - m_ * cos(theta)) * (m_ * g_ * sin(theta) - m_ * l_ * sin(theta) * omega ** 2 + F[0])
    model.set_dynamical_equations([v, dv, omega, 1. / l_ * (dv * cos(theta) + g_ * sin(theta)), F[1]])
    dt = .1
    model.setup(dt=dt)
    x0 = [2.5, 0., 1.5, 0., 0.]

    def build(custom):
        nmpc = NMPC(model)
        nmpc.quad_stage_cost.add_states(names=['theta', 'dummy'], ref=[np.pi, 10], weights=[np.pi, 10])
        nmpc.horizon = 10
        nmpc.set_box_constraints(x_ub=[3, 0.5, 10, 10, 10000], x_lb=[2, -0.5, -10, -10, 0])
        nmpc.set_initial_guess(x_guess=x0, u_guess=[0., 0.])
        if custom:
            nmpc.set_custom_constraints_function(lambda v, xi, ui: _trapezoid(4, dt)(v, xi, ui), ub=4, lb=0)
        nmpc.setup()
        return nmpc

    def integral(nmpc):
        x_opt, _, _ = nmpc.return_prediction()
        return float(((x_opt[0, 4, :-1] + x_opt[0, 4, 1:]) / 2 * dt).sum())
    free = build(False)
    free.optimize(x0)
    assert free.solver_status_code[0] == 1 and integral(free) > 4.2
    # the same row SOFT under the default transcription (mpc.py:1551-1556, :1731-1740): the slack is the last entry of v, two rows at
    # the end of g; the penalty 1e4 e^2 lets the integral exceed 4 by e, and stationarity in e reads 2e4 e = the upper row's multiplier
    soft = NMPC(model)
    soft.quad_stage_cost.add_states(names=['theta', 'dummy'], ref=[np.pi, 10], weights=[np.pi, 10])
    soft.horizon = 10
    soft.set_box_constraints(x_ub=[3, 0.5, 10, 10, 10000], x_lb=[2, -0.5, -10, -10, 0])
    soft.set_initial_guess(x_guess=x0, u_guess=[0., 0.])
    soft.set_custom_constraints_function(lambda v, xi, ui: _trapezoid(4, dt)(v, xi, ui), ub=4, lb=0, soft=True, max_violation=1.)
    soft.setup(solver_options={'tol': 1e-10})
    assert (soft._n_v, soft._n_g) == (free._n_v + 1, free._n_g + 2) and soft._e_cus_ind == [soft._n_v - 1]
    soft.optimize(x0)
    assert soft.solver_status_code[0] == 1
    vs, ls = soft._nlp_solution['x'].cpu().numpy()[0], soft._nlp_solution['lam_g'].cpu().numpy()[0]
    e = vs[-1]
    assert 1e-6 < e < 1. and abs(integral(soft) - e - 4.) < 1e-6
    np.testing.assert_allclose(ls[-2], 2e4 * e, rtol=1e-5)
    assert abs(ls[-1]) < 1e-6
    nmpc = build(True)
    assert nmpc._nlp_options['integration_method'] == 'collocation' and nmpc._n_g == free._n_g + 1 and nmpc._n_v == free._n_v
    nmpc.optimize(x0)
    assert nmpc.solver_status_code[0] == 1
    assert integral(nmpc) - 4 < 1e-3 and abs(integral(nmpc) - 4) < 1e-6           # the reference's assertion; active
    lam = nmpc._nlp_solution['lam_g'].cpu().numpy()
    assert lam[0, -1] > 1e-3
    assert float(nmpc._nlp_solution['f'][0]) > float(soft._nlp_solution['f'][0]) > float(free._nlp_solution['f'][0])


def test_collocation_custom_row_in_a_batch_and_across_warm_started_calls():
    """The same problem for THREE instances and two consecutive calls: the output pass of the collocation transcription addresses
    the engine's compact rows and the rows of `v` per instance - both carry the accumulator of the custom row as a hidden tail entry
    - and the second call warm-starts from `v`, whose hidden entry pins the accumulator's start value (round-5 advisor finding: with
    the tail left out of the row lengths, every instance behind the first read shifted rows, and the second call started the
    accumulator from uninitialised memory).  Every instance must reproduce its own single-instance solve, in both calls."""
    from hilo_mpc_amd import NMPC, Model
    from hilo_mpc_amd.expr import cos, sin
    M, m_, l_, g_ = 5., 1., 1., 9.81
    model = Model()
    x = model.set_dynamical_states(['x', 'v', 'theta', 'omega', 'dummy'])
    F = model.set_inputs(['F', 'u_dummy'])
    v, theta, omega = x[1], x[2], x[3]
    dv = 1. / (M + m_ - m_ * cos(theta)) * (m_ * g_ * sin(theta) - m_ * l_ * sin(theta) * omega ** 2 + F[0])
    model.set_dynamical_equations([v, dv, omega, 1. / l_ * (dv * cos(theta) + g_ * sin(theta)), F[1]])
    dt = .1
    model.setup(dt=dt)
    X0 = np.array([[2.5, 0., 1.5, 0., 0.], [2.4, .1, 1.4, .1, 0.], [2.6, -.1, 1.6, -.1, 0.]])

    def build():
        nmpc = NMPC(model)
        nmpc.quad_stage_cost.add_states(names=['theta', 'dummy'], ref=[np.pi, 10], weights=[np.pi, 10])
        nmpc.horizon = 10
        nmpc.set_box_constraints(x_ub=[3, 0.5, 10, 10, 10000], x_lb=[2, -0.5, -10, -10, 0])
        nmpc.set_initial_guess(x_guess=list(X0[0]), u_guess=[0., 0.])
        nmpc.set_custom_constraints_function(lambda v, xi, ui: _trapezoid(4, dt)(v, xi, ui), ub=4, lb=0)
        nmpc.setup(solver_options={'ipopt.tol': 1e-11})     # (two solves that stop at 1e-8 agree to ~5e-6 only)
        return nmpc

    def integral(nmpc):
        x_opt, _, _ = nmpc.return_prediction()
        return ((x_opt[:, 4, :-1] + x_opt[:, 4, 1:]) / 2 * dt).sum(axis=1)
    single = []
    for b in range(3):
        one = build()
        one.optimize(X0[b])
        assert one.solver_status_code[0] == 1
        single.append((one._nlp_solution['x'].cpu().numpy()[0], one._nlp_solution['lam_g'].cpu().numpy()[0]))
    nmpc = build()
    for call in range(2):
        nmpc.optimize(X0)
        assert np.all(nmpc.solver_status_code == 1)
        np.testing.assert_allclose(integral(nmpc), 4., atol=1e-6)
        vb, lb = nmpc._nlp_solution['x'].cpu().numpy(), nmpc._nlp_solution['lam_g'].cpu().numpy()
        assert np.all(np.isfinite(vb)) and np.all(np.isfinite(lb))
        for b in range(3):
            assert np.max(np.abs(vb[b] - single[b][0]) / np.maximum(1., np.abs(single[b][0]))) < 1e-6, (call, b)
            assert np.max(np.abs(lb[b] - single[b][1]) / np.maximum(1., np.abs(single[b][1]))) < 2e-5, (call, b)
    assert np.all(nmpc._nlp_solution['iter_count'].cpu().numpy() <= 12)            # the second call started at the solution
