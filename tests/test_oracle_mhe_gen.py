"""The general MHE oracle (oracle/mhe_gen.py) against the three earlier oracles on the cases those cover, its hard stage
constraint rows against scipy SLSQP on the same NLP, and the layout figures the product must reproduce (CPU only)."""
import numpy as np
import pytest

from oracle import models
from oracle.mhe import MheEstIpm, MheEstProblem, MheIpm, MheProblem
from oracle.mhe_coll import MheCollIpm, MheCollProblem
from oracle.mhe_gen import MheGenIpm, MheGenProblem
from oracle.nmpc import IpmOptions
from tests.problems import C3B, c3_data

N, B = 4, 2
O = IpmOptions(tol=1e-10)


def _kw():
    kw = {k: v for k, v in C3B.items() if k not in ('model', 'p')}
    kw.update(N=N, order=2)                                  # (the symbolic second derivatives of the RK4 map take a minute to build)
    return kw, {k: v for k, v in kw.items() if k != 'order'}


def test_reproduces_the_discrete_oracles():
    m = models.get('chemostat4')
    xa, um, ym, _ = c3_data(B, N=N)
    kw, kg = _kw()
    r0 = MheIpm(MheProblem(m, **kw), O).solve(xa, C3B['p'], um, ym)
    r1 = MheGenIpm(MheGenProblem(m, degree=0, order=2, **kg), O).solve(xa, [], C3B['p'], um, ym)
    assert np.array_equal(r0['iters'], r1['iters'])
    np.testing.assert_allclose(r1['v'], r0['v'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r1['lam'], r0['lam'], rtol=0, atol=1e-12)
    e = dict(est=[2], Wp=[1e-2], p_lb=[.2], p_ub=[2.], p_guess=[.7])
    r0 = MheEstIpm(MheEstProblem(m, **e, **kw), O).solve(xa, [.7], [100., 4., 0.], um, ym)
    r1 = MheGenIpm(MheGenProblem(m, degree=0, order=2, **e, **kg), O).solve(xa, [.7], [100., 4., 0.], um, ym)
    assert np.array_equal(r0['iters'], r1['iters'])
    np.testing.assert_allclose(r1['v'], r0['v'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r1['p_opt'], r0['p_opt'], rtol=0, atol=1e-12)


def test_reproduces_the_collocation_oracle():
    m = models.get('chemostat4')
    xa, um, ym, _ = c3_data(B, N=N)
    _, kg = _kw()
    pbc = MheCollProblem(m, degree=3, **kg)
    r0 = MheCollIpm(pbc, O).solve(xa, C3B['p'], um, ym)
    pb = MheGenProblem(m, degree=3, **kg)
    r1 = MheGenIpm(pb, O).solve(xa, [], C3B['p'], um, ym)
    assert (pb.n_v, pb.n_g) == (pbc.n_v, pbc.n_g) and np.array_equal(r0['iters'], r1['iters'])
    np.testing.assert_allclose(r1['v'], r0['v'], rtol=0, atol=1e-10)
    np.testing.assert_allclose(r1['lam'], r0['lam'], rtol=0, atol=1e-10)


@pytest.mark.parametrize('degree,noise', [(2, False), (0, True)])
def test_hard_stage_constraint_rows_vs_slsqp(degree, noise):
    """mhe.py:536-553, :749-757: rows at the collocation points and at the nodes k < N.  The interior-point solution satisfies the
    rows, is active somewhere, and scipy SLSQP on the same variables / rows / objective arrives at the same objective."""
    m = models.get('chemostat4')
    xa, um, ym, _ = c3_data(1, N=N)
    _, kg = _kw()
    kg = {q: v for q, v in kg.items() if noise or q not in ('Ww', 'w_lb', 'w_ub')}
    free = MheGenIpm(MheGenProblem(m, degree=degree, order=2, noise=noise, **kg), O).solve(xa, [], C3B['p'], um, ym)
    ub = float(free['X'][:, :N, 0].max() * .97)
    cons = dict(expr=['X', 'P + 2*I*X'], lb=[-np.inf, 0.], ub=[ub, np.inf])
    pb = MheGenProblem(m, degree=degree, order=2, noise=noise, constraint=cons, **kg)
    ipm = MheGenIpm(pb, O)
    r = ipm.solve(xa, [], C3B['p'], um, ym)
    assert r['status'][0] == 1 and r['f'][0] > free['f'][0] * 1.01
    assert pb.n_g == N * (degree * 4 + 4 + (degree + 1) * 2) and ipm.lam_g(r).shape == (1, pb.n_g)
    top = max(r['X'][0, :N, 0].max(), r['Xc'][0, ..., 0].max() if degree else -1.)
    assert abs(top - ub) < 1e-7
    lam = ipm.lam_g(r).reshape(N, -1)
    rows = np.concatenate([lam[:, :2 * degree], lam[:, 2 * degree + degree * 4 + 4:]], axis=1)
    assert rows[:, 0::2].max() > 1e-2 and rows[:, 0::2].min() > -1e-9 and np.abs(rows[:, 1::2]).max() < 1e-8   # upper bound: lam >= 0
    # stationarity by central differences of the oracle's own f and rows (independent of its analytic derivatives): the only
    # non-zero components of grad f + J' lam are the bound multipliers, of the right sign at active bounds
    data = ipm.data(xa, [], C3B['p'], um, ym)
    w = r['w'][0]
    fc = lambda q: ipm.eval_fc(q[None], data)                                       # noqa: E731
    g = np.zeros(ipm.nw)
    h = 1e-6
    for i in range(ipm.nw):
        e = np.zeros(ipm.nw)
        e[i] = h * max(1., abs(w[i]))
        (fp, cp), (fm, cm) = fc(w + e), fc(w - e)
        g[i] = ((fp[0] - fm[0]) + r['lam'][0] @ (cp[0] - cm[0])) / (2 * e[i])
    at_lb, at_ub = w - ipm.lb < 1e-7, ipm.ub - w < 1e-7
    assert np.abs(g[~(at_lb | at_ub)]).max() < 1e-6
    assert np.all(g[at_lb] > -1e-6) and np.all(g[at_ub] < 1e-6) and np.abs(fc(w)[1]).max() < 1e-9
    if degree:                                                                       # (the finite-difference SLSQP run takes minutes with noise)
        from scipy.optimize import minimize
        w0 = w + 1e-3 * np.random.default_rng(0).standard_normal(ipm.nw) * np.maximum(1e-2, np.abs(w))
        bnds = [(None if not np.isfinite(a) else a, None if not np.isfinite(b) else b) for a, b in zip(ipm.lb, ipm.ub)]
        s = minimize(lambda q: float(fc(q)[0][0]), w0, method='SLSQP', bounds=bnds, constraints=[dict(type='eq', fun=lambda q: fc(q)[1][0])],
                     options=dict(maxiter=300, ftol=1e-14))
        assert s.status in (0, 8) and abs(s.fun - r['f'][0]) < 1e-7 * max(1., abs(r['f'][0]))
