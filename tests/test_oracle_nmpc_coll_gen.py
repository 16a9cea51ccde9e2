"""The oracle of the general collocation transcription (oracle/nmpc_coll_gen.py: path variable, algebraic states, stage
constraints at the collocation points and the node) checked on the CPU: it reproduces the two pinned / established oracles it
generalises iteration by iteration, its derivatives equal finite differences, and an independent solver (scipy SLSQP) finds the
same minimiser of the same NLP."""
import numpy as np

from oracle import models
from oracle.nmpc import IpmOptions
from oracle.nmpc_coll import CollIpm, CollNmpcProblem
from oracle.nmpc_coll_gen import GenCollIpm, GenCollProblem
from oracle.nmpc_dae import DaeCollIpm, DaeCollProblem
from tests.problems import C2, C5DS, c2_x0, c5_x0, oracle_coll_gen


def test_reduces_to_the_plain_collocation_oracle():
    spec = dict(C2, N=4)
    kw = {k: v for k, v in spec.items() if k not in ('model', 'p', 'order')}
    x0 = c2_x0(2)
    for obj in ('continuous', 'discrete'):
        pb0 = CollNmpcProblem(models.get('chemostat4'), objective=obj, **kw)
        r0 = CollIpm(pb0, IpmOptions(tol=1e-10)).solve(x0, spec['p'])
        pb1 = GenCollProblem(models.get('chemostat4'), objective=obj, **kw)
        ipm = GenCollIpm(pb1, IpmOptions(tol=1e-10))
        r1 = ipm.solve(x0, spec['p'])
        assert np.array_equal(r0['iters'], r1['iters']) and np.all(r1['status'] == 1)
        np.testing.assert_allclose(ipm.to_v(r1), CollIpm(pb0).to_v(r0), rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(r1['lam'], r0['lam'], rtol=1e-7, atol=1e-9)
        assert (pb1.n_v, pb1.n_g) == (pb0.n_v, pb0.n_g)


def test_reduces_to_the_dae_oracle():
    kw = dict(dt=.1, N=4, z_guess=[1.4], stage_states=[([1, 2], [10., 5.], [0., 0.])], stage_inputs=[([0], [.1], None)],
              x_lb=[-5, -10, -10, -10], x_ub=[5, 10, 10, 10], x_guess=[2.5, 0., .1, 0.], u_guess=[0.])
    x0 = np.array([[2.5, 0., .1, 0.], [2., .2, -.1, .1]])
    pb0 = DaeCollProblem(models.get('pendulum4_dae'), **kw)
    i0 = DaeCollIpm(pb0, IpmOptions(tol=1e-10))
    r0 = i0.solve(x0, [])
    pb1 = GenCollProblem(models.get('pendulum4_dae'), **kw)
    i1 = GenCollIpm(pb1, IpmOptions(tol=1e-10))
    r1 = i1.solve(x0, [])
    assert np.array_equal(r0['iters'], r1['iters']) and np.all(r1['status'] == 1)
    np.testing.assert_allclose(i1.to_v(r1), i0.to_v(r0), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(r1['lam'], r0['lam'], rtol=1e-10, atol=1e-12)
    assert (pb1.n_v, pb1.n_g, pb1.z_ind, pb1.ip_ind, pb1.zp_ind) == (pb0.n_v, pb0.n_g, pb0.z_ind, pb0.ip_ind, pb0.zp_ind)


def _c5d(N, degree):
    return dict(C5DS, N=N, collocation=dict(degree=degree))


def test_layout_of_configuration_5_on_the_dae():
    """mpc.py:1462-1548: [x | u | z nodes | (ip_k, zp_k) | e]; rows per interval: d x 2 soft rows, d x (7 + 1) collocation rows,
    7 continuity rows, 2 soft rows at the node."""
    pb = oracle_coll_gen(_c5d(5, 3))
    assert pb.n_v == 6 * 7 + 5 * 3 + 6 * 1 + 5 * (21 + 3) + 1 and pb.n_g == 5 * (6 + 24 + 7 + 2)
    assert pb.z_ind[0] == [57] and pb.ip_ind[0] == list(range(63, 84)) and pb.zp_ind[0] == [84, 85, 86] and pb.e_ind == [183]
    assert pb.nrow == 1 and pb.rows[0][5] == 0                  # only the upper row of the pair is bounded


def test_derivatives_equal_finite_differences_and_slsqp_agrees():
    from scipy.optimize import minimize
    for degree in (2, 1):
        pb = oracle_coll_gen(_c5d(3, degree))
        ipm = GenCollIpm(pb, IpmOptions(tol=1e-10))
        r = ipm.solve(c5_x0(2), [])
        assert np.all(r['status'] == 1) and r['kkt'].max() < 1e-9
        w, lam = r['w'][:1].copy(), r['lam'][:1]
        data = {'x0': r['x0'][:1], 'p': np.zeros((1, 0))}
        f, g, c, J, W = ipm.eval_all(w, lam, data)
        h = 1e-6
        gn, Jn, Wn = np.zeros_like(g), np.zeros_like(J), np.zeros_like(W)
        for i in range(ipm.nw):
            wp, wm = w.copy(), w.copy()
            wp[0, i] += h
            wm[0, i] -= h
            fp, cp = ipm.eval_fc(wp, data)
            fm, cm = ipm.eval_fc(wm, data)
            gn[0, i], Jn[0, :, i] = (fp - fm)[0] / (2 * h), (cp - cm)[0] / (2 * h)
            _, gp_, _, Jp, _ = ipm.eval_all(wp, lam, data)
            _, gm_, _, Jm, _ = ipm.eval_all(wm, lam, data)
            Wn[0, :, i] = ((gp_ + np.einsum('bmi,bm->bi', Jp, lam)) - (gm_ + np.einsum('bmi,bm->bi', Jm, lam)))[0] / (2 * h)
        assert np.abs(g - gn).max() < 1e-6 * max(1., np.abs(g).max())
        assert np.abs(J - Jn).max() < 1e-7 * max(1., np.abs(J).max())
        assert np.abs(W - Wn).max() < 1e-5 * max(1., np.abs(W).max())
        # an independent solver on the same NLP from the guess: same objective and minimiser
        w0 = ipm._with_slacks(ipm.start(data['x0'], data), data)[0]
        lb, ub = ipm.lb.copy(), ipm.ub.copy()
        res = minimize(lambda q: ipm.eval_fc(q[None], data)[0][0], w0, jac=lambda q: ipm.eval_all(q[None], lam * 0, data)[1][0],
                       method='SLSQP', bounds=list(zip(np.where(np.isfinite(lb), lb, None), np.where(np.isfinite(ub), ub, None))),
                       constraints=[{'type': 'eq', 'fun': lambda q: ipm.eval_fc(q[None], data)[1][0],
                                     'jac': lambda q: ipm.eval_all(q[None], lam * 0, data)[3][0]}],
                       options={'ftol': 1e-13, 'maxiter': 500})
        assert res.status in (0, 8), res.message       # 8: no descent left at round-off level (the comparison below decides)
        np.testing.assert_allclose(res.fun, r['f'][0], rtol=1e-7)
        if degree > 1:
            # (degree 1 with the continuous objective: theta_0 and u_theta,0 only enter through theta_0 + dt u_theta,0 - the
            # minimiser is a segment, an interior-point method lands on its analytic centre, an SQP method anywhere on it)
            np.testing.assert_allclose(res.x[:ipm.o_s], w[0, :ipm.o_s], rtol=1e-4, atol=2e-5)


def test_hard_terminal_constraint_rows():
    """mpc.py:1693-1700 under collocation: hard rows on the end state of the last interval, in g between its continuity rows and
    its node rows.  (1) with bounds that do not bind the solution is the unconstrained one; (2) with binding bounds the rows hold,
    the stationarity of the oracle's own NLP holds by central differences of its f and rows (independent of its analytic
    derivatives), and the multiplier of the last continuity rows carries grad c_t' nu_t in the reference's convention."""
    spec = dict(C2, N=4)
    kw = {k: v for k, v in spec.items() if k not in ('model', 'p', 'order')}
    m = models.get('chemostat4')
    x0 = c2_x0(2)
    o = IpmOptions(tol=1e-10)
    free = GenCollIpm(GenCollProblem(m, degree=2, **kw), o)
    r0 = free.solve(x0, spec['p'])
    xN = r0['X'][:, -1] * free.pb.sx
    loose = GenCollIpm(GenCollProblem(m, degree=2, terminal=dict(expr=['I*S', 'S'], lb=[-1e3, -np.inf], ub=[np.inf, 1e6]), **kw), o)
    r1 = loose.solve(x0, spec['p'])
    np.testing.assert_allclose(loose.to_v(r1), free.to_v(r0), rtol=1e-8, atol=1e-10)
    # binding: an EQUALITY on the product I * S (the mean of the free solutions' values) and the substrate at the end above the free
    # solutions' - both within reach of the two feeds
    prod, smin = float((xN[:, 3] * xN[:, 1]).mean()), float(xN[:, 1].max() * 1.002)
    pb = GenCollProblem(m, degree=2, terminal=dict(expr=['I*S', 'S'], lb=[prod, smin], ub=[prod, np.inf]), **kw)
    ipm = GenCollIpm(pb, o)
    r = ipm.solve(x0, spec['p'])
    assert np.all(r['status'] == 1) and pb.n_g == free.pb.n_g + 2 and ipm.lam_g(r).shape == (2, pb.n_g)
    xe = r['X'][:, -1] * pb.sx
    assert np.all(xe[:, 1] >= smin - 1e-6) and np.all(np.abs(xe[:, 3] * xe[:, 1] - prod) < 1e-6 * prod) and np.all(r['f'] > r0['f'])
    data = {'x0': r['x0'], 'p': r['p_data']}
    for b in range(2):
        w = r['w'][b]
        fc = lambda q: ipm.eval_fc(q[None], {k: v[b:b + 1] for k, v in data.items()})          # noqa: E731
        g = np.zeros(ipm.nw)
        for i in range(ipm.nw):
            e = np.zeros(ipm.nw)
            e[i] = 1e-6 * max(1., abs(w[i]))
            (fp, cp), (fm, cm) = fc(w + e), fc(w - e)
            g[i] = ((fp[0] - fm[0]) + r['lam'][b] @ (cp[0] - cm[0])) / (2 * e[i])
        at = (w - ipm.lb < 1e-6) | (ipm.ub - w < 1e-6)
        assert np.abs(g[~at]).max() < 2e-6 and np.abs(fc(w)[1]).max() < 1e-8
    lg = ipm.lam_g(r).reshape(2, -1)
    per = (pb.n_g - 2) // pb.N
    lt = lg[:, (pb.N - 1) * per + 2 * 4 + 4:(pb.N - 1) * per + 2 * 4 + 4 + 2]                    # [coll rows (d nx) | continuity (nx) | terminal]
    assert np.all(np.abs(lt[:, 0]) > 1e-4)                                                       # the equality carries force


def test_minimum_time_variables_and_rows():
    """mpc.py:859-866, :1606-1617, :1746-1754 restated natively: N sampling-interval variables (last block of v, >= 0, guess dt),
    N - 1 rows dt_k - dt_{k+1} = 0 at the end of g, J = weight * sum(dt) - on the race car of tests/test_NMPC.py:2706-2760 with
    N = 8.  The layout figures, equal intervals, the terminal equality, the speed limit at every collocation point, and the
    stationarity of the oracle's own f and rows by central differences (incl. the dt columns)."""
    N = 8
    pb = GenCollProblem(models.get('racecar2'), dt=.1, N=N, degree=3, constraint=dict(expr=['v - (1 - sin(2*pi*p)/2)'], lb=[-np.inf], ub=[0.]),
                        terminal=dict(expr=['p'], lb=[1.], ub=[1.]), min_time=1., u_lb=[0.], u_ub=[1.], x_guess=[0., 0.], u_guess=[0.])
    assert pb.n_v == (N + 1) * 2 + N + N * 3 * 2 + N and pb.dt_ind == list(range(pb.n_v - N, pb.n_v))
    assert pb.n_g == N * (3 * 1 + 3 * 2 + 2 + 1) + 1 + (N - 1)
    ipm = GenCollIpm(pb, IpmOptions(tol=1e-10))
    r = ipm.solve(np.array([[0., 0.]]), [])
    assert r['status'][0] == 1
    dt = r['dt'][0]
    assert np.allclose(dt, dt[0], rtol=0, atol=1e-12) and abs(r['f'][0] - dt.sum()) < 1e-12 and 1.8 < dt.sum() < 2.1
    assert abs(r['X'][0, -1, 0] - 1.) < 2e-8                                         # the equality row: met inside the relaxed box
    Xc = r['Xc'][0]
    assert np.all(Xc[..., 1] - (1 - np.sin(2 * np.pi * Xc[..., 0]) / 2) < 1e-7)
    v, lg = ipm.to_v(r), ipm.lam_g(r)
    assert v.shape == (1, pb.n_v) and lg.shape == (1, pb.n_g) and np.allclose(v[0, pb.dt_ind], dt)
    data = {'x0': r['x0'], 'p': r['p_data']}
    w = r['w'][0]
    fc = lambda q: ipm.eval_fc(q[None], data)                                          # noqa: E731
    g = np.zeros(ipm.nw)
    for i in range(ipm.nw):
        e = np.zeros(ipm.nw)
        e[i] = 1e-6 * max(1., abs(w[i]))
        (fp, cp), (fm, cm) = fc(w + e), fc(w - e)
        g[i] = ((fp[0] - fm[0]) + r['lam'][0] @ (cp[0] - cm[0])) / (2 * e[i])
    at = (w - ipm.lb < 1e-6) | (ipm.ub - w < 1e-6)
    assert np.abs(g[~at]).max() < 5e-6 and np.abs(fc(w)[1]).max() < 1e-8
    # the multipliers of the dt rows telescope: eta_k - eta_{k-1} = -(weight + mu_k' f) - nonzero, and the last one closes the sum
    eta = lg[0, -(N - 1):]
    assert np.abs(eta).max() > 1e-3
