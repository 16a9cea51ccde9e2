"""oracle/mhe_coll.py (collocation inside the moving-horizon estimator, the reference's default, mhe.py:512-593): the solve ends
at a KKT point of its own transcription, and - two discretisations of the same estimation problem - agrees with the
explicit-integrator transcription of oracle/mhe.py to the order of the integration error."""
import numpy as np

from oracle import models
from oracle.mhe import MheIpm, MheProblem
from oracle.mhe_coll import MheCollIpm, MheCollProblem
from oracle.nmpc import IpmOptions
from tests.problems import C3B, c3_data


def _spec(N):
    return {k: v for k, v in dict(C3B, N=N).items() if k not in ('model', 'p', 'order')}


def test_collocation_mhe_solves_to_a_kkt_point_and_matches_rk4():
    N, B = 6, 3
    xa, um, ym, _ = c3_data(B, N=N)
    pb = MheCollProblem(models.get('chemostat4'), degree=3, **_spec(N))
    assert pb.n_v == 4 + (N + 1) * 4 + N * 4 + N * 3 * 4 and pb.n_g == N * (3 * 4 + 4)
    assert pb.ip_ind[0][0] == 4 + (N + 1) * 4 + N * 4 and pb.ip_ind[-1][-1] == pb.n_v - 1
    ipm = MheCollIpm(pb, IpmOptions(tol=1e-10))
    res = ipm.solve(xa, C3B['p'], um, ym)
    assert np.all(res['status'] == 1)
    assert res['v'].shape == (B, pb.n_v)
    # KKT residual of the returned point, recomputed from eval_all
    data = {'p': np.tile(C3B['p'], (B, 1)), 'x_arrival': xa, 'u_meas': um, 'y_meas': ym}
    f, g, c, J, W = ipm.eval_all(res['w'], res['lam'], data)
    assert np.abs(c).max() < 1e-9
    r = g + np.einsum('bmw,bm->bw', J, res['lam']) - res['zl'] + res['zu']
    assert np.abs(r).max() < 1e-7
    # the same window with RK4: Radau-3 (order 5) and RK4 (order 4) at dt = 0.25 differ by the integration error only
    ref = MheIpm(MheProblem(models.get('chemostat4'), order=4, **_spec(N)), IpmOptions(tol=1e-10)).solve(xa, C3B['p'], um, ym)
    assert np.all(ref['status'] == 1)
    np.testing.assert_allclose(res['x_opt'], ref['x_opt'], rtol=2e-3, atol=1e-4)
    np.testing.assert_allclose(res['f'], ref['f'], rtol=5e-2)


def test_first_derivatives_of_the_collocation_transcription_by_finite_differences():
    N, B = 3, 1
    xa, um, ym, _ = c3_data(B, N=N)
    pb = MheCollProblem(models.get('chemostat4'), degree=2, **_spec(N))
    ipm = MheCollIpm(pb)
    rng = np.random.default_rng(1)
    w = np.concatenate([np.tile(pb.x_guess, N + 1), np.zeros(N * 4), np.tile(pb.x_guess, N * 2)])[None, :] * \
        (1 + .05 * rng.normal(size=(1, ipm.nw))) + 1e-3 * rng.normal(size=(1, ipm.nw))
    data = {'p': np.atleast_2d(C3B['p']), 'x_arrival': xa, 'u_meas': um, 'y_meas': ym}
    lam = rng.normal(size=(1, ipm.m))
    f, g, c, J, W = ipm.eval_all(w, lam, data)
    f0, c0 = ipm.eval_fc(w, data)
    np.testing.assert_allclose(f, f0, rtol=1e-13)
    np.testing.assert_allclose(c, c0, rtol=1e-13, atol=1e-15)
    h = 1e-6
    for j in range(ipm.nw):
        wp, wm = w.copy(), w.copy()
        wp[0, j] += h
        wm[0, j] -= h
        fp, cp = ipm.eval_fc(wp, data)
        fm, cm = ipm.eval_fc(wm, data)
        np.testing.assert_allclose((fp - fm) / (2 * h), g[:, j], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose((cp - cm)[0] / (2 * h), J[0, :, j], rtol=2e-5, atol=2e-6)
        # Hessian of the Lagrangian column j by differences of the gradient of L = f + lam^T c
        _, gp, _, Jp, _ = ipm.eval_all(wp, lam, data)
        _, gm, _, Jm, _ = ipm.eval_all(wm, lam, data)
        Lp = gp + np.einsum('bmw,bm->bw', Jp, lam)
        Lm = gm + np.einsum('bmw,bm->bw', Jm, lam)
        np.testing.assert_allclose((Lp - Lm)[0] / (2 * h), W[0, :, j], rtol=5e-4, atol=5e-4 * max(1., np.abs(W[0]).max()) * 1e-2)
