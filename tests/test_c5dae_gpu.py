"""GPU parity of BASELINE configuration 5 as it is written: path-following NMPC on a DAE with soft constraints - and of what it is
made of: the reference's default transcription (collocation) together with a path variable (mpc.py:1173-1204), algebraic states
(mpc.py:1488-1527) and nonlinear stage constraints, which the reference imposes at every collocation point as well as at the node
(mpc.py:1338-1356, :1700-1725).  The product eliminates the collocation and the algebraic states inside the shooting map and
rebuilds them - and the multipliers of their rows - in the reference's layout; the oracle (oracle/nmpc_coll_gen.py) carries them
as variables like the reference.  Both run at tol = 1e-9 (two solvers that stop at a KKT error of 1e-8 agree to 5e-5
only; 1e-10 is below what the penalty terms of 1e4 leave of the scaled error); tolerances: v 1e-6 relative, f 1e-9, u0 1e-6,
multipliers 1e-5."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.nmpc import IpmOptions                                                                  # noqa: E402
from oracle.nmpc_coll_gen import GenCollIpm                                                          # noqa: E402
from tests.problems import C2, C5D, C5DS, c2_x0, c5_x0, oracle_coll_gen, product_gen                 # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'nmpc_c5dae.json')
TOL = 1e-9


def _layout(nmpc, pb):
    assert (nmpc._n_v, nmpc._n_g) == (pb.n_v, pb.n_g)
    assert nmpc._x_ind == pb.x_ind and nmpc._u_ind == pb.u_ind and nmpc._e_soft_stage_ind == pb.e_ind
    assert nmpc._ip_ind == pb.ip_ind and nmpc._z_ind == pb.z_ind and nmpc._zp_ind == pb.zp_ind


def _close(nmpc, ipm, ref, u, sel, vtol, ltol):
    v, vr = nmpc._nlp_solution['x'].cpu().numpy()[sel], ipm.to_v(ref)[sel]
    assert np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))) < vtol
    np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy()[sel], ref['f'][sel], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(u[sel], ref['u0'][sel], rtol=vtol, atol=vtol)
    lam, lr = nmpc._nlp_solution['lam_g'].cpu().numpy()[sel], ipm.lam_g(ref)[sel]
    assert np.max(np.abs(lam - lr) / np.maximum(1., np.abs(lr))) < ltol


def _compare(spec, x0, p, vtol=1e-6, ltol=1e-5, other_minimum=0):
    """EVERY instance must reproduce the oracle's point - primal, objective, multipliers - when it starts there: the oracle's KKT
    point is the product's.  Then from the reference's guess: the path cost makes the NLP non-convex, and the product (collocation
    and algebraic states eliminated) and the oracle (all of them variables) walk different iteration paths that may end in
    different local minima - at most `other_minimum` instances may (their objectives then differ visibly), the others agree."""
    pb = oracle_coll_gen(spec)
    ipm = GenCollIpm(pb, IpmOptions(tol=TOL))
    ref = ipm.solve(x0, p)
    nmpc = product_gen(spec, **{'ipopt.tol': TOL})
    _layout(nmpc, pb)
    cp = p if len(p) else None
    assert np.all(ref['status'] == 1)
    u = nmpc.optimize(x0, cp=cp, v0=ipm.to_v(ref))
    assert np.all(nmpc.solver_status_code == 1)
    _close(nmpc, ipm, ref, u, np.ones(len(ref['f']), dtype=bool), vtol, ltol)
    nmpc.reset_solution()
    u = nmpc.optimize(x0, cp=cp)
    assert np.array_equal(nmpc.solver_status_code, ref['status'])
    f = nmpc._nlp_solution['f'].cpu().numpy()
    same = np.abs(f - ref['f']) <= 1e-8 * np.maximum(1., np.abs(ref['f']))
    assert (~same).sum() <= other_minimum, (f, ref['f'])
    _close(nmpc, ipm, ref, u, same, vtol, ltol)
    return nmpc, pb, ipm, ref


def test_c5_dae_short_horizon_vs_oracle():
    """N = 10, B = 8: theta inside the collocation scheme, the soft limit z <= 4 on the ALGEBRAIC state at the three collocation
    points and the node of every interval, one shared slack behind the collocation blocks of v."""
    x0 = c5_x0(8)
    nmpc, pb, ipm, ref = _compare(C5DS, x0, [], other_minimum=1)
    assert (pb.n_v, pb.n_g) == (11 * 7 + 10 * 3 + 11 + 10 * 24 + 1, 10 * (3 * 2 + 24 + 7 + 2))
    assert ref['E'].max() > 1e-3                                                        # some x_0 break the limit: the slack is used
    e = nmpc.stage_constraint.e_soft_value.cpu().numpy()
    np.testing.assert_allclose(e, ref['E'], rtol=1e-6, atol=1e-8)
    v = nmpc._nlp_solution['x'].cpu().numpy()
    lam = nmpc._nlp_solution['lam_g'].cpu().numpy().reshape(8, pb.N, -1)
    # per interval [3 x (z - e <= 4, dropped lower row) | 3 x (7 ode rows, 1 algebraic row) | 7 continuity | node rows (2)]
    assert np.all(lam[:, :, [1, 3, 5, 38]] == 0.)                                      # rows without a finite bound: dropped
    assert lam[:, :, [0, 2, 4, 37]].max() > 1e-3 and lam[:, :, [0, 2, 4, 37]].min() >= 0.   # active upper rows: multipliers >= 0
    # the algebraic equation holds at the collocation points of the returned vector
    for k in range(pb.N):
        Xc = v[:, pb.ip_ind[k]].reshape(8, 3, 7)
        Zc = v[:, pb.zp_ind[k]].reshape(8, 3)
        np.testing.assert_allclose(Zc, Xc[:, :, 1] ** 2 + Xc[:, :, 3] ** 2, rtol=1e-12, atol=1e-12)
    # closed loop, warm start (the reference re-uses the previous solution, mpc.py:725-726): both continue from the ORACLE's point
    x1 = nmpc.plant_step(x0, ref['u0']).cpu().numpy()
    ref2 = ipm.solve(x1, [], w0=ipm.w_from_v(ipm.to_v(ref)))
    u2 = nmpc.optimize(x1, v0=ipm.to_v(ref))
    assert np.array_equal(nmpc.solver_status_code, ref2['status'])
    same = np.abs(nmpc._nlp_solution['f'].cpu().numpy() - ref2['f']) <= 1e-8 * np.maximum(1., np.abs(ref2['f']))
    assert same.sum() >= 5
    np.testing.assert_allclose(u2[same], ref2['u0'][same], rtol=1e-6, atol=1e-6)


def test_rows_active_at_the_collocation_points_algebraic_rows_carry_force():
    """A speed band 2.5 <= z + 0.1 a <= 4 (soft, both rows of the pair) on the algebraic state and the acceleration: the lower row is
    active at the collocation points of most intervals, so the multipliers of the algebraic rows are of order 0.1 - the part of
    `lam_g` the plain configuration leaves at zero.  (The input inside the expression keeps the reference's NLP regular: a row that
    only sees states is imposed twice on the same point - at the last Radau point of an interval and at the next node - and its two
    multipliers are then only determined as a sum.)"""
    spec = dict(C5DS, constraint=dict(expr=['z + 0.1 * a'], lb=[2.5], ub=[4.], soft=True))
    nmpc, pb, ipm, ref = _compare(spec, c5_x0(6), [], other_minimum=1)
    lam = ipm.lam_g(ref).reshape(6, pb.N, -1)
    assert pb.n_g == 10 * (3 * 2 + 24 + 7 + 2)
    assert np.abs(lam[:, :, [6 + 7, 6 + 15, 6 + 23]]).max() > 1e-2 and np.abs(lam[:, :, [1, 3, 5]]).max() > 1e-2


def test_degree_one_node_rows_see_the_collocation_points_algebraic_state():
    """degree = 1 is the only degree for which the reference's node residual `_stage_constraints_fun(.., zp[ii, 0], ..)`
    (mpc.py:1707) type-checks with algebraic states: it receives z of the single collocation point (Radau: the END of the
    interval).  Restated as it is, on both sides."""
    spec = dict(C5DS, N=6, collocation=dict(degree=1))
    _compare(spec, c5_x0(4), [], other_minimum=2)      # (implicit Euler + quadrature at the end point: a flat, multi-modal landscape)
    band = dict(spec, constraint=dict(expr=['z + 0.1 * a'], lb=[2.5], ub=[4.], soft=True))
    _compare(band, c5_x0(4), [], other_minimum=2)


def test_hard_and_two_sided_rows_on_an_ode_under_collocation():
    """chemostat4 (ODE) under collocation: hard rows (one of them one-sided from below), soft two-sided rows, the discrete objective
    with degree 2.  The expressions involve the inputs: a row of states only, active at the last Radau point of an interval AND at
    the next node (the same point), has multipliers that are only determined as a sum in the reference's NLP - the product and the
    oracle split it differently (measured: 4e-4 on ONE pair of rows, everything else equal to 1e-9)."""
    kw = dict(C2, N=8, collocation=dict(degree=3))
    kw.pop('order', None)
    hard = dict(kw, constraint=dict(expr=['X * S + 20 * DS', 'S - X + 5 * DI'], lb=[-np.inf, 0.], ub=[60., np.inf]))
    nmpc, pb, ipm, ref = _compare(hard, c2_x0(4), C2['p'])
    assert pb.n_g == 8 * (3 * 2 + 12 + 4 + 2)
    lam = ipm.lam_g(ref).reshape(4, pb.N, -1)
    assert np.abs(lam[:, :, :6]).max() > 1e-3                                   # rows at collocation points are active
    soft = dict(kw, constraint=dict(expr=['X * S + 20 * DS'], lb=[2.], ub=[60.], soft=True, weight=[[1e3]], max_violation=[5.]))
    _compare(soft, c2_x0(4), C2['p'])
    # Gauss-Legendre points: no collocation point coincides with a node, every row is imposed once (with Radau points and an input
    # at its bound the pair of rows of an interval's end point is degenerate again)
    disc = dict(hard, collocation=dict(degree=2, objective='discrete', points='legendre'))
    _compare(disc, c2_x0(4), C2['p'])


@pytest.mark.parametrize('case', ['c5ds', 'band', 'degree1', 'degree1_band', 'ode_hard', 'ode_soft', 'legendre2_discrete'])
def test_workspace_mode_lanes_solve_the_collocation_systems_together(case, monkeypatch):
    """The long horizons keep the iterate in a global-memory workspace, and there the lanes of a wave factor an interval's collocation
    system together and the derivatives come from the implicit-function theorem (hilo_nmpc_user.h::coll_pass) - a different code
    path from the one the short problems above take.  HILO_FORCE_BIG sends the same short problems down that path, against the same
    oracle: every lane layout (right-hand sides as second columns: 21 x 21, 12 x 12, 8 x 8 systems; in lanes of their own: degree 1),
    algebraic states, rows at the collocation points, hard and soft, continuous and discrete objective."""
    monkeypatch.setenv('HILO_FORCE_BIG', '1')
    kw = dict(C2, N=8, collocation=dict(degree=3))
    kw.pop('order', None)
    hard = dict(kw, constraint=dict(expr=['X * S + 20 * DS', 'S - X + 5 * DI'], lb=[-np.inf, 0.], ub=[60., np.inf]))
    if case == 'c5ds':
        _compare(C5DS, c5_x0(8), [], other_minimum=1)
    elif case == 'band':
        _compare(dict(C5DS, constraint=dict(expr=['z + 0.1 * a'], lb=[2.5], ub=[4.], soft=True)), c5_x0(6), [], other_minimum=1)
    elif case == 'degree1':
        _compare(dict(C5DS, N=6, collocation=dict(degree=1)), c5_x0(4), [], other_minimum=2)
    elif case == 'degree1_band':
        _compare(dict(C5DS, N=6, collocation=dict(degree=1), constraint=dict(expr=['z + 0.1 * a'], lb=[2.5], ub=[4.], soft=True)),
                 c5_x0(4), [], other_minimum=2)
    elif case == 'ode_hard':
        _compare(hard, c2_x0(4), C2['p'])
    elif case == 'ode_soft':
        _compare(dict(kw, constraint=dict(expr=['X * S + 20 * DS'], lb=[2.], ub=[60.], soft=True, weight=[[1e3]], max_violation=[5.])),
                 c2_x0(4), C2['p'])
    else:
        _compare(dict(hard, collocation=dict(degree=2, objective='discrete', points='legendre')), c2_x0(4), C2['p'])


def test_c5_dae_full_horizon_vs_fixture_and_batch_properties():
    """N = 50 (the configuration's horizon), B = 3 against the oracle's solution stored by tests/golden/make_c5dae_golden.py (the
    dense oracle needs minutes for it); then B = 1024 in closed loop: every instance solved, the slack covers the limit at every
    node and collocation point, z is consistent with the states."""
    with open(GOLD) as f:
        g = json.load(f)
    x0 = np.array(g['x0'])
    assert all(s in (1, 2) for s in g['status']) and max(g['kkt']) < 1e-9       # (the oracle ran at tol 1e-10: one instance stops at 2.7e-10)
    nmpc = product_gen(C5D, **{'ipopt.tol': 1e-10})        # the fixture's tolerance (at 1e-9 the long horizon's flat directions leave 9e-6)
    assert (nmpc._n_v, nmpc._n_g) == (g['n_v'], g['n_g']) == (51 * 7 + 50 * 3 + 51 + 50 * 24 + 1, 50 * 39)
    vr, lr, fr, ur = np.array(g['v']), np.array(g['lam_g']), np.array(g['f']), np.array(g['u0'])

    def close(sel):
        v = nmpc._nlp_solution['x'].cpu().numpy()[sel]
        assert np.max(np.abs(v - vr[sel]) / np.maximum(1., np.abs(vr[sel]))) < 1e-6
        np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy()[sel], fr[sel], rtol=1e-9)
        lam = nmpc._nlp_solution['lam_g'].cpu().numpy()[sel]
        assert np.max(np.abs(lam - lr[sel]) / np.maximum(1., np.abs(lr[sel]))) < 1e-5
    # from the oracle's point: every instance stays there (its KKT point is the product's); from the guess: the non-convex path
    # cost may send an instance to another local minimum (tests above) - at most one of the three
    u = nmpc.optimize(x0, v0=vr)
    assert np.all(nmpc.solver_status_code == 1)
    close(np.ones(3, dtype=bool))
    np.testing.assert_allclose(u, ur, rtol=1e-6, atol=1e-6)
    nmpc.reset_solution()
    u = nmpc.optimize(x0)
    assert np.all(nmpc.solver_status_code == 1)
    same = np.abs(nmpc._nlp_solution['f'].cpu().numpy() - fr) <= 1e-8 * np.maximum(1., np.abs(fr))
    assert same.sum() >= 2
    close(same)
    np.testing.assert_allclose(u[same], ur[same], rtol=1e-6, atol=1e-6)
    nm = product_gen(C5D)
    x = c5_x0(1024)
    for _ in range(3):
        u = nm.optimize(x)
        assert np.all(nm.solver_status_code == 1), np.unique(nm.solver_status_code, return_counts=True)
        v = nm._nlp_solution['x'].cpu().numpy()
        e = v[:, nm._e_soft_stage_ind]
        assert np.all(e >= -1e-8)
        for k in range(50):
            Xc = v[:, nm._ip_ind[k]].reshape(-1, 3, 7)
            Zc = v[:, nm._zp_ind[k]].reshape(-1, 3)
            assert np.abs(Zc - (Xc[:, :, 1] ** 2 + Xc[:, :, 3] ** 2)).max() < 1e-10
            assert np.all(Zc <= 4. + e + 1e-6)
            xk = v[:, nm._x_ind[k]]
            assert np.all(xk[:, 1] ** 2 + xk[:, 3] ** 2 <= 4. + e[:, 0] + 1e-6)
        x = nm.plant_step(x, u).cpu().numpy()
    assert np.all(np.isfinite(x))
