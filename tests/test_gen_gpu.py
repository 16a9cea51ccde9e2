"""GPU parity of the general NMPC path (SURVEY 8 rows a4 path following, a5 GenericConstraint; configuration C5):
`hilo_nmpc_solve` through the reference-style API vs the oracle's dense interior-point solver on the reference's
transcription (oracle/nmpc_gen.py).  Tolerances as in tests/test_nmpc_gpu.py: status codes exact, primal 5e-5 at the
default tolerance 1e-8, objective 1e-8."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=['precompiled', 'runtime'])
def backend(request, monkeypatch):
    """Every case on both routes: the library's precompiled variants (expression interpreter) and the general policy compiled at
    run time, which is the default for problems with a path variable or nonlinear constraints (hilo_mpc_amd/nmpc.py)."""
    monkeypatch.setenv('HILO_NMPC_BACKEND', request.param)
    return request.param

from oracle.nmpc_gen import GenIpm                                                              # noqa: E402
from tests.problems import C2, C2H, C2S, C5, C5S, c2_x0, c5_x0, oracle_gen, product_gen         # noqa: E402


def _compare(spec, x0, p, vtol=5e-5, itol=6):
    pb = oracle_gen(spec)
    ipm = GenIpm(pb)
    ref = ipm.solve(x0, p)
    nmpc = product_gen(spec)
    assert (nmpc._n_v, nmpc._n_g) == (pb.n_v, pb.n_g)
    assert nmpc._x_ind == pb.x_ind and nmpc._u_ind == pb.u_ind and nmpc._e_soft_stage_ind == pb.e_ind
    u = nmpc.optimize(x0, cp=p if len(p) else None)
    assert np.array_equal(nmpc.solver_status_code, ref['status'])
    assert np.all(ref['status'] == 1)
    v, vr = nmpc._nlp_solution['x'].cpu().numpy(), ipm.to_v(ref)
    assert np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))) < vtol
    np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy(), ref['f'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(u, ref['u0'], rtol=vtol, atol=vtol)
    assert np.all(nmpc.stats()['kkt_error'] <= 1e-8)
    # iteration counts are a property of the algorithm, not of parity: same order only (round-off steers the inertia
    # correction and the filter differently in the dense and the Riccati factorisation)
    assert np.max(np.abs(nmpc.stats()['iter_count'] - ref['iters'])) <= itol
    return nmpc, pb, ipm, ref


def test_hard_constraint_vs_oracle():
    nmpc, pb, ipm, ref = _compare(C2H, c2_x0(8), C2['p'])
    XS = ref['X'][:, :-1, 0] * ref['X'][:, :-1, 1]
    assert XS.max() > 59.9                                            # the limit X*S <= 60 is active
    xp, up, _ = nmpc.return_prediction()
    assert (xp[:, 0, :-1] * xp[:, 1, :-1]).max() <= 60. + 1e-4
    # multipliers in the reference's g order [defect | constraint row] per stage
    lam = ipm.lam_g(ref)
    lam = lam.reshape(lam.shape[0], pb.N, -1)
    lam[:, -1, :pb.nxa] += 2 * (ref['X'][:, -1] - pb.xrefNa) @ pb.WNa  # terminal cost on Phi_{N-1} (mpc.py:1682)
    np.testing.assert_allclose(nmpc._nlp_solution['lam_g'].cpu().numpy(), lam.reshape(lam.shape[0], -1), rtol=2e-4, atol=1e-5)


def test_soft_constraint_vs_oracle():
    nmpc, pb, ipm, ref = _compare(C2S, c2_x0(8), C2['p'])
    e = nmpc.stage_constraint.e_soft_value.cpu().numpy()
    np.testing.assert_allclose(e, ref['E'], atol=1e-6)
    assert np.all(e >= -1e-8)


def test_soft_constraint_two_sided_rows():
    """lb and ub finite: both rows c - e <= ub and -c - e <= -lb (mpc.py:1276-1277), 2 multipliers per stage."""
    spec = dict(C2, constraint=dict(expr=['X * S'], lb=[2.], ub=[60.], soft=True, weight=[[1e3]], max_violation=[5.]))
    nmpc, pb, ipm, ref = _compare(spec, c2_x0(4), C2['p'])
    assert nmpc._n_g == pb.N * (pb.nx + 2)


def test_path_following_soft_speed_limit_vs_oracle():
    """C5 at N = 10: theta_0 free, virtual input, path cost nonlinear in theta, soft |v|^2 <= 4 with x_0 at the limit."""
    x0 = c5_x0(8)
    nmpc, pb, ipm, ref = _compare(C5S, x0, [])
    assert ref['E'].max() > 1e-3                                      # some x_0 violate the limit: the slack is used
    xp, up, _ = nmpc.return_prediction()
    assert xp.shape == (8, 7, 11) and up.shape == (8, 3, 10)
    np.testing.assert_allclose(xp[:, :6, 0], x0, rtol=1e-12)
    assert np.all(np.diff(xp[:, 6], axis=1) > 0)                      # theta advances (u_theta >= 1e-4)
    # closed loop: warm start from the previous solution, as the reference does
    x1 = nmpc.plant_step(x0, ref['u0']).cpu().numpy()
    ref2 = ipm.solve(x1, [], w0=ipm.w_from_v(ipm.to_v(ref)))
    u2 = nmpc.optimize(x1)
    assert np.array_equal(nmpc.solver_status_code, ref2['status'])
    np.testing.assert_allclose(u2, ref2['u0'], rtol=5e-5, atol=5e-5)


def test_constraint_on_the_path_variable_vs_oracle():
    """A stage constraint that involves the path variable - a soft tube |px - sin(theta)| <= 0.15 around the first path coordinate
    next to the soft speed limit (two slacks): theta is a state of the augmented model (mpc.py:1181-1191), the expression is
    compiled into the run-time compiled policy (the precompiled variants' interpreter has no slot for it)."""
    spec = dict(C5S, constraint=dict(expr=['vx**2 + vy**2', 'px - sin(theta)'], lb=[-np.inf, -0.15], ub=[4., 0.15], soft=True))
    x0 = c5_x0(4)
    nmpc, pb, ipm, ref = _compare(spec, x0, [], itol=10)
    assert nmpc._jit and (nmpc._n_v, nmpc._n_g) == (109, 110)
    xp, _, _ = nmpc.return_prediction()
    e = nmpc.stage_constraint.e_soft_value.cpu().numpy()
    assert np.abs(xp[:, 0] - np.sin(xp[:, 6])).max() > 0.149 and e[:, 1].max() > 1e-5      # the tube is active, its slack used
    assert np.all(np.abs(xp[:, 0, :-1] - np.sin(xp[:, 6, :-1])) <= 0.15 + e[:, 1:2] + 1e-6)


def test_c5_full_horizon_global_workspace():
    """C5 as configured (N = 50: 8 engine states, 3 inputs): the iterate does not fit the 160 KB of LDS and lives in the
    per-instance global workspace; same algorithm, same parity bar.  Then the BASELINE batch per GPU (1024) in closed loop."""
    x0 = c5_x0(3)
    nmpc, pb, ipm, ref = _compare(C5, x0, [], itol=15)
    assert (nmpc._n_v, nmpc._n_g) == (51 * 7 + 50 * 3 + 1, 50 * (7 + 2))      # theta-augmented nx=7, nu=3
    x = c5_x0(1024)
    for _ in range(3):
        u = nmpc.optimize(x)
        x = nmpc.plant_step(x, u).cpu().numpy()
    st = nmpc.solver_status_code
    assert np.mean(st == 1) >= 0.99
    xp, up, _ = nmpc.return_prediction()
    assert np.all(xp[:, 1] ** 2 + xp[:, 3] ** 2 <= 4. + nmpc.stage_constraint.e_soft_value.cpu().numpy() + 1e-6)


def test_path_following_without_constraint():
    spec = {k: v for k, v in C5S.items() if k != 'constraint'}
    _compare(spec, c5_x0(4), [])


def test_general_errors():
    from hilo_mpc_amd import NMPC, Model
    from hilo_mpc_amd._lib import HiloError
    m = Model('pendulum4').discretize('rk4').setup(dt=.1)
    nmpc = NMPC(m)
    nmpc.quad_stage_cost.add_states(names=['v'], weights=[1.])
    nmpc.horizon = 5
    nmpc.stage_constraint.constraint = m.x['v'] ** 2
    nmpc.stage_constraint.ub = [4.]
    # no precompiled variant for (pendulum4, one hard row): compiled at run time for the zoo functor instead of failing
    nmpc.setup(options={'integration_method': 'discrete'})
    assert nmpc._jit
    u = nmpc.optimize(np.array([[.2, 1., .1, 0.]]))
    assert nmpc.solver_status_code[0] == 1 and np.all(np.isfinite(u))
    with pytest.raises(TypeError, match="is_soft must be of type bool"):
        nmpc.stage_constraint.is_soft = 1
    deep = m.x['v']
    for _ in range(9):
        deep = 1. / (1. + deep * deep)
    e = m.x['v']
    for _ in range(9):
        e = m.x['x'] + (m.x['v'] * e)                               # right-nested: needs a deeper stack each level
    with pytest.raises(ValueError, match="too deep"):
        e.program()


@pytest.mark.parametrize('name', ['soft+scaling', 'path+soft+scaling'])
def test_general_problems_with_scaling(name):
    """Scaling together with constraints / path following: the constraint acts on un-scaled quantities (modeling.py:843-849),
    the quadratic cost on scaled ones (modeling.py:310)."""
    if name == 'soft+scaling':
        _compare(dict(C2S, x_scaling=[.1, 40., 2., 1.], u_scaling=[2., 2.]), c2_x0(4), C2['p'])
    else:
        _compare(dict(C5S, x_scaling=[1., 2., 1., 2., 1., 1.], u_scaling=[5., 5.]), c5_x0(4), [])


def test_hard_terminal_constraint_vs_oracle():
    """`nmpc.terminal_constraint` (hard): rows on the integrated end state, between the last defect and the last stage rows
    in g (mpc.py:1693-1700); here together with a hard stage constraint."""
    tc = dict(expr=['X + P', 'S'], lb=[-np.inf, 30.], ub=[1.0, np.inf])
    for spec in (dict(C2, N=8, terminal_constraint=tc), dict(C2H, N=8, terminal_constraint=dict(expr=['X + P'], lb=[-np.inf], ub=[1.]))):
        nmpc, pb, ipm, ref = _compare(spec, c2_x0(4), C2['p'])
        xp, up, _ = nmpc.return_prediction()
        assert np.all(xp[:, 0, -1] + xp[:, 2, -1] <= 1. + 1e-5)
        lam = ipm.lam_g(ref)
        nxa = pb.nxa
        last = (pb.N - 1) * (nxa + pb.n_con_ref)
        lam[:, last:last + nxa] += 2 * (ref['X'][:, -1] - pb.xrefNa) @ pb.WNa          # terminal cost convention (mpc.py:1682)
        np.testing.assert_allclose(nmpc._nlp_solution['lam_g'].cpu().numpy(), lam, rtol=2e-4, atol=2e-5)


def test_free_initial_state_vs_oracle():
    """`optimize(x0, fix_x0=False)` (mpc.py:797-807): x_0 is a variable inside the state box, the measured state is ignored;
    switching back to fix_x0=True restores the classic problem on the same handle."""
    from tests.problems import product_nmpc, oracle_problem
    from oracle.nmpc import DenseIpm
    spec = dict(C2, N=8, x_lb=[1., 10., 0., 0.], x_ub=[8., 60., 5., 20.], x_guess=[4., 30., 1., 5.])
    pb = oracle_gen(spec)
    ipm = GenIpm(pb, free_x0=True)
    x0 = c2_x0(4)
    ref = ipm.solve(x0, C2['p'])
    assert np.all(ref['status'] == 1)
    nmpc = product_nmpc(spec)
    u = nmpc.optimize(x0, cp=C2['p'], fix_x0=False)
    assert np.array_equal(nmpc.solver_status_code, ref['status'])
    v, vr = nmpc._nlp_solution['x'].cpu().numpy(), ipm.to_v(ref)
    assert np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))) < 5e-5
    np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy(), ref['f'], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(u, ref['u0'], rtol=5e-5, atol=5e-5)
    assert np.abs(v[:, :4] - x0 / pb.sx).max() > 1e-2                      # x_0 really moved away from the measurement
    fixed = DenseIpm(oracle_problem(spec)).solve(x0, C2['p'])
    nmpc2 = product_nmpc(spec)
    nmpc2.optimize(x0, cp=C2['p'], fix_x0=False)
    nmpc2._nlp_options['warm_start'] = False
    u2 = nmpc2.optimize(x0, cp=C2['p'])                                     # fix_x0=True again, cold start
    assert np.array_equal(nmpc2.solver_status_code, fixed['status'])
    np.testing.assert_allclose(u2, fixed['u0'], rtol=5e-5, atol=5e-5)
    # own box for x_0 (optimize(fix_x0=False, x0_lb=, x0_ub=), mpc.py:803-807), then back to the state box on the same handle
    box = ([2., 20., 0., 1.], [3., 35., 2., 8.])
    refb = GenIpm(pb, free_x0=True, x0_box=box).solve(x0, C2['p'])
    assert np.all(refb['status'] == 1)
    nmpc._nlp_options['warm_start'] = False
    ub_ = nmpc.optimize(x0, cp=C2['p'], fix_x0=False, x0_lb=box[0], x0_ub=box[1])
    vb = nmpc._nlp_solution['x'].cpu().numpy()
    assert np.array_equal(nmpc.solver_status_code, refb['status'])
    assert np.max(np.abs(vb - ipm.to_v(refb)) / np.maximum(1., np.abs(ipm.to_v(refb)))) < 5e-5
    np.testing.assert_allclose(ub_, refb['u0'], rtol=5e-5, atol=5e-5)
    assert np.all(vb[:, :4] >= np.array(box[0]) - 1e-6) and np.all(vb[:, :4] <= np.array(box[1]) + 1e-6)
    u3 = nmpc.optimize(x0, cp=C2['p'], fix_x0=False)
    np.testing.assert_allclose(u3, ref['u0'], rtol=5e-5, atol=5e-5)


def test_soft_terminal_constraint_vs_oracle():
    """`set_terminal_constraints(..., is_soft=True)` (mpc.py:1684-1692): rows on x_{N-1} with the slack e_soft_term behind
    the stage slack in v, its penalty once in the objective; alone and next to a hard stage constraint."""
    for spec in (dict(C2, N=8, terminal_constraint=dict(expr=['X + P'], lb=[0.5], ub=[1.0], soft=True, weight=[[50.]])),
                 dict(C2H, N=8, terminal_constraint=dict(expr=['S'], lb=[45.], ub=[np.inf], soft=True, max_violation=[2.]))):
        pb = oracle_gen(spec)
        ipm = GenIpm(pb)
        x0 = c2_x0(4)
        ref = ipm.solve(x0, C2['p'])
        assert np.all(ref['status'] == 1)
        nmpc = product_gen(spec)
        assert (nmpc._n_v, nmpc._n_g) == (pb.n_v, pb.n_g) and nmpc._e_soft_term_ind == pb.eT_ind
        u = nmpc.optimize(x0, cp=C2['p'])
        assert np.array_equal(nmpc.solver_status_code, ref['status'])
        v, vr = nmpc._nlp_solution['x'].cpu().numpy(), ipm.to_v(ref)
        assert np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))) < 5e-5
        np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy(), ref['f'], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(u, ref['u0'], rtol=5e-5, atol=5e-5)
        np.testing.assert_allclose(nmpc.terminal_constraint.e_soft_value.cpu().numpy()[:, 0], vr[:, pb.eT_ind[0]], atol=5e-5)
        lam = ipm.lam_g(ref)
        last = (pb.N - 1) * (pb.nxa + pb.n_con_ref)
        lam[:, last:last + pb.nxa] += 2 * (ref['X'][:, -1] - pb.xrefNa) @ pb.WNa        # terminal cost convention (mpc.py:1682)
        np.testing.assert_allclose(nmpc._nlp_solution['lam_g'].cpu().numpy(), lam, rtol=2e-4, atol=2e-5)
    # soft stage AND soft terminal constraint: two slacks - no precompiled variant, compiled at run time for the zoo functor
    both = dict(C2S, N=8, terminal_constraint=dict(expr=['S'], lb=[45.], ub=[np.inf], soft=True))
    nmpc, pb, ipm, ref = _compare(both, c2_x0(4), C2['p'])
    assert nmpc._jit and nmpc._e_soft_term_ind == pb.eT_ind and len(pb.e_ind) == 1
    # slack VECTORS: two soft stage rows and two soft terminal expressions (one of them two-sided), e_soft_stage and e_soft_term
    # side by side behind the inputs in v (mpc.py:1529-1548), a non-default terminal weight
    vec = dict(C2, N=8, constraint=dict(expr=['X * S', 'P'], lb=[-np.inf, -np.inf], ub=[60., 1.2], soft=True),
               terminal_constraint=dict(expr=['S', 'X + P'], lb=[45., 0.5], ub=[np.inf, 1.0], soft=True, weight=[[50., 0.], [0., 80.]]))
    nmpc, pb, ipm, ref = _compare(vec, c2_x0(4), C2['p'])
    assert nmpc._e_soft_stage_ind == pb.e_ind == [52, 53] and nmpc._e_soft_term_ind == pb.eT_ind == [54, 55]
    vr = ipm.to_v(ref)
    np.testing.assert_allclose(nmpc.terminal_constraint.e_soft_value.cpu().numpy(), vr[:, pb.eT_ind], atol=5e-5)
    assert np.all(vr[:, pb.eT_ind[1]] > .1)                 # the second terminal slack is active


def test_c5_from_a_cold_zero_velocity_guess_solves_everywhere():
    """C5 (N = 50, soft speed limit, iterate in the global workspace) started from an all-zero state guess - far from the path,
    the line search has to go through the feasibility restoration - and three closed-loop steps: every instance status 1."""
    import torch
    spec = dict(C5, x_guess=[0., 0., 0., 0., 0., 0.])
    nmpc = product_gen(spec)
    x = torch.as_tensor(c5_x0(256), device='cuda')
    for _ in range(3):
        u = nmpc.optimize(x)
        assert np.all(nmpc.solver_status_code == 1), np.unique(nmpc.solver_status_code, return_counts=True)
        x = nmpc.plant_step(x, u)

