"""CPU: the collocation oracle (oracle/nmpc_coll.py).  The polynomial basis is pinned against the published Radau IIA
tableau (Hairer & Wanner, order 5) and the product's own basis builder; the derivatives the dense solver uses are checked
by finite differences; the solve is checked by its KKT residual and against an RK4 solve of the same problem."""
import numpy as np

from oracle import models
from oracle.nmpc import DenseIpm
from oracle.nmpc_coll import CollIpm, CollNmpcProblem, collocation_points, polynomial_basis
from tests.problems import C2, c2_x0, oracle_problem


def test_radau_basis_is_radau_iia():
    s6 = np.sqrt(6.)
    np.testing.assert_allclose(collocation_points(3), [(4 - s6) / 10, (4 + s6) / 10, 1.], rtol=1e-14)
    B, C, D, tau = polynomial_basis(3)
    np.testing.assert_allclose(B, [0., (16 - s6) / 36, (16 + s6) / 36, 1. / 9], atol=1e-14)
    np.testing.assert_allclose(D, [0., 0., 0., 1.], atol=1e-13)
    A = np.linalg.inv(C[1:, 1:].T)
    A_ref = np.array([[(88 - 7 * s6) / 360, (296 - 169 * s6) / 1800, (-2 + 3 * s6) / 225],
                      [(296 + 169 * s6) / 1800, (88 + 7 * s6) / 360, (-2 - 3 * s6) / 225],
                      [(16 - s6) / 36, (16 + s6) / 36, 1. / 9]])
    np.testing.assert_allclose(A, A_ref, atol=1e-13)
    np.testing.assert_allclose(C.sum(axis=0), 0., atol=1e-12)              # derivative of the constant vanishes
    from hilo_mpc_amd.nmpc import _collocation_basis                         # the product's own restatement
    pb = _collocation_basis(3, 'radau')
    np.testing.assert_allclose(pb['A'], A_ref, atol=1e-13)
    np.testing.assert_allclose(pb['D'], D, atol=1e-13)
    np.testing.assert_allclose(collocation_points(2, 'legendre'), [.5 - np.sqrt(3) / 6, .5 + np.sqrt(3) / 6], rtol=1e-14)


def _problem(N=4):
    kw = {k: v for k, v in C2.items() if k not in ('model', 'p', 'order')}
    kw['N'] = N
    pb = CollNmpcProblem(models.get('chemostat4'), **kw)
    return pb, CollIpm(pb)


def test_layout_and_derivatives():
    pb, ipm = _problem(4)
    assert pb.n_v == 5 * 4 + 4 * 2 + 4 * 12 and pb.n_g == 4 * 16          # mpc.py:1440-1443, :1657-1669
    assert pb.ip_ind[0][0] == 28 and pb.ip_ind[3][-1] == pb.n_v - 1
    rng = np.random.default_rng(0)
    x0 = c2_x0(1)
    data = {'x0': x0 / pb.sx, 'p': np.atleast_2d(C2['p'])}
    w = np.concatenate([np.tile(pb.x_guess, pb.N), np.tile(pb.u_guess + .2, pb.N), np.tile(pb.x_guess, pb.N * 3)])[None]
    w = w * (1 + .05 * rng.uniform(-1, 1, w.shape)) + .01
    lam = rng.normal(size=(1, ipm.m))
    f, g, c, J, W = ipm.eval_all(w, lam, data)
    eps = 1e-6
    gn, Jn, Wn = np.zeros_like(g), np.zeros_like(J), np.zeros_like(W)
    for i in range(ipm.nw):
        wp, wm = w.copy(), w.copy()
        wp[0, i] += eps
        wm[0, i] -= eps
        fp, cp = ipm.eval_fc(wp, data)
        fm, cm = ipm.eval_fc(wm, data)
        gn[0, i] = (fp - fm)[0] / (2 * eps)
        Jn[0, :, i] = (cp - cm)[0] / (2 * eps)
        _, gp, _, Jp, _ = ipm.eval_all(wp, lam, data)
        _, gm, _, Jm, _ = ipm.eval_all(wm, lam, data)
        Wn[0, :, i] = ((gp + np.einsum('bmi,bm->bi', Jp, lam)) - (gm + np.einsum('bmi,bm->bi', Jm, lam)))[0] / (2 * eps)
    assert np.abs(gn - g).max() < 1e-6 and np.abs(Jn - J).max() < 1e-6 and np.abs(Wn - W).max() < 1e-5


def test_collocation_solve_close_to_rk4():
    pb, ipm = _problem(6)
    x0 = c2_x0(2)
    res = ipm.solve(x0, C2['p'])
    assert np.all(res['status'] == 1) and np.all(res['kkt'] <= 1e-8)
    rk = DenseIpm(oracle_problem(dict(C2, N=6))).solve(x0, C2['p'])
    assert 1e-6 < np.abs(res['U'] - rk['U']).max() < 5e-2                 # same problem, two discretisations of order 5 / 4
    assert np.abs(res['f'] - rk['f']).max() < 1.


def test_cstr_notebook_numbers_are_reproduced_by_the_oracle():
    """The reference's only published NMPC result, docs/docsource/examples/CSTR_Example.ipynb cell 16:
    'True: Q: 59882.1817 C_A: 0.4912 C_B: 0.5088 T: 438.4732' after 1000 closed-loop steps.  The committed fixture
    (tests/golden/make_cstr_golden.py) holds the oracle's closed loop; its last line equals the notebook's to every printed
    digit, and re-running the last ten steps from the stored state and warm start with the code as it is now lands on the
    fixture's final point - the fixture is what this oracle computes, not a hand-edited number."""
    import json
    import os
    from tests.problems import CSTR_PRINTED, cstr_oracle, cstr_plant
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'nmpc_cstr.json')))
    want = 'Q: {} C_A: {} C_B: {} T: {}'.format(*CSTR_PRINTED)
    assert fx['printed_by_the_reference'] == want and fx['printed_by_the_oracle'] == want
    pb, ipm = cstr_oracle()
    x, w = np.array([fx['restart']['x']]), np.array([fx['restart']['w']])
    for _ in range(1000 - fx['restart']['step']):
        res = ipm.solve(x, [], w0=w)
        assert res['status'][0] == 1
        w = res['w']
        x = cstr_plant(x, res['u0'])
    line = f"Q: {res['u0'][0, 0]:.4f} C_A: {x[0, 0]:.4f} C_B: {x[0, 1]:.4f} T: {x[0, 2]:.4f}"
    assert line == want
    np.testing.assert_allclose(x[0], fx['final']['x'], rtol=1e-12)
    assert abs(res['u0'][0, 0] - fx['final']['u']) < 1e-6
    # the transient is part of the pin: the loop is at the steady state long before step 1000
    assert abs(fx['snapshots'][5]['x'][2] - fx['final']['x'][2]) < 1e-3
