"""oracle/mhe.py (moving-horizon estimator with state noise on a pre-discretised model, mhe.py:596-790) - parity unpinned: the
reference holds no numbers for it, so the oracle is held (a) to the reference's integer bookkeeping, (b) to an independent solver
(scipy SLSQP) on the same estimation problem written down again from the model equations - own right-hand side, own Runge-Kutta
step, own objective - (c) to finite differences for every derivative the interior-point solver uses, (d) to the invariance the
scaling has to have."""
import numpy as np
from scipy.optimize import minimize

from oracle.mhe import MheIpm
from oracle.nmpc import IpmOptions
from tests.problems import C3B, c3_data, oracle_mhe


def test_bookkeeping_of_the_decision_vector():
    """mhe.py:614-655: v = [p | x_0..x_N | w_0..w_{N-1}], N nx rows (mhe.py:733-740)."""
    pb = oracle_mhe(dict(C3B, N=5))
    assert pb.n_v == 4 + 6 * 4 + 5 * 4 and pb.n_g == 5 * 4
    assert pb.p_ind == [[0, 1, 2, 3]]
    assert pb.x_ind[0] == [4, 5, 6, 7] and pb.x_ind[5] == [24, 25, 26, 27]
    assert pb.w_ind[0] == [28, 29, 30, 31] and pb.w_ind[4] == [44, 45, 46, 47]


def _ode(x, u, p):
    # `ecoli_D1210_conti('simple')` with the closed-form rates (hilo_mpc/library/models.py:163-198, :143-148), written again
    X, S, P, I = x
    DS, DI = u
    Sf, If, ISF, IRF = p
    phi = 0.407 * S / (0.108 + S + S ** 2 / 14814.0)
    mu = phi * (ISF + 0.22 * IRF / (0.22 + I))
    D = DS + DI
    return np.array([mu * X - D * X, -2 * mu * X - D * S + DS * Sf, phi * (0.0005 + I) / (0.022 + I) * X - D * P, -D * I + DI * If])


def _rk4(x, u, p, h):
    k1 = _ode(x, u, p)
    k2 = _ode(x + h / 2 * k1, u, p)
    k3 = _ode(x + h / 2 * k2, u, p)
    k4 = _ode(x + h * k3, u, p)
    return x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)


def test_estimate_vs_slsqp_on_the_problem_written_again():
    """arrival term on x_0, measurement and noise terms for k = 1..N-1 (no stage cost at k = 0, mhe.py:742-748), rows
    x_{k+1} = Phi(x_k, u_k) + w_k, boxes on x and w - minimised by SLSQP with finite-difference gradients, which pins the objective
    to ~1e-5 relative and the well observable states to ~1e-6; S and I are weakly observable."""
    N = 4
    spec = dict(C3B, N=N)
    xa, um, ym, _ = c3_data(2, N=N, seed=9)
    ref = MheIpm(oracle_mhe(spec), IpmOptions(tol=1e-10)).solve(xa, spec['p'], um, ym)
    assert np.all(ref['status'] == 1)
    p = np.array(spec['p'])
    for b in range(2):
        def split(w):
            return w[:(N + 1) * 4].reshape(N + 1, 4), w[(N + 1) * 4:].reshape(N, 4)

        def obj(w):
            X, W = split(w)
            d = X[0] - xa[b]
            f = 4. * d @ d
            for k in range(1, N):
                r = X[k][[0, 2]] - ym[b, k]
                f += 16. * r @ r + 1e6 * W[k] @ W[k]
            return f

        def eq(w):
            X, W = split(w)
            return np.concatenate([X[k + 1] - (_rk4(X[k], um[b, k], p, spec['dt']) + W[k]) for k in range(N)])
        lb = np.concatenate([np.zeros((N + 1) * 4), np.full(N * 4, -1e-3)])
        ub = np.concatenate([np.full((N + 1) * 4, np.inf), np.full(N * 4, 1e-3)])
        w0 = np.concatenate([np.tile(spec['x_guess'], N + 1), np.zeros(N * 4)])
        sol = minimize(obj, w0, method='SLSQP', bounds=list(zip(lb, ub)), constraints=[{'type': 'eq', 'fun': eq}],
                       options={'ftol': 1e-14, 'maxiter': 1000})
        X, W = split(sol.x)
        assert np.abs(eq(sol.x)).max() < 1e-10
        assert sol.fun >= ref['f'][b] * (1 - 1e-9) and abs(sol.fun - ref['f'][b]) < 2e-5 * ref['f'][b]
        np.testing.assert_allclose(X[:, [0, 2]], ref['X'][b][:, [0, 2]], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(X, ref['X'][b], rtol=1e-4, atol=2e-3)
        np.testing.assert_allclose(W[1:], ref['Wn'][b][1:], atol=2e-5)      # w_0 carries no cost: it trades against x_0 inside its box
        # the oracle's point is feasible and at least as good for the problem written here
        wr = np.concatenate([ref['X'][b].ravel(), ref['Wn'][b].ravel()])
        assert np.abs(eq(wr)).max() < 1e-9 and abs(obj(wr) - ref['f'][b]) < 1e-12 + 1e-10 * ref['f'][b]


def test_derivatives_by_finite_differences():
    """gradient, constraint Jacobian and Hessian of the Lagrangian of `MheIpm.eval_all` at a random interior point."""
    spec = dict(C3B, N=3, x_scaling=[.5, 20., .1, .2], w_scaling=[1e-3, 1e-2, 1e-3, 1e-3], u_scaling=[.1, .05])
    pb = oracle_mhe(spec)
    ipm = MheIpm(pb)
    xa, um, ym, xt = c3_data(1, N=3, seed=2)
    rng = np.random.default_rng(0)
    data = {'p': np.atleast_2d(spec['p']), 'x_arrival': xa, 'u_meas': um / pb.su, 'y_meas': ym}
    w = np.concatenate([(xt[0] / pb.sx).ravel() * (1 + .05 * rng.normal(size=16)), .1 * rng.normal(size=12)])[None]
    lam = rng.normal(size=(1, ipm.m))
    f, g, c, J, W = ipm.eval_all(w, lam, data)
    f0, c0 = ipm.eval_fc(w, data)
    assert abs(f0[0] - f[0]) < 1e-12 * max(1., abs(f[0])) and np.abs(c0 - c).max() < 1e-13
    h = 1e-6

    def lag_grad(wv):
        _, gg, _, JJ, _ = ipm.eval_all(wv, lam, data)
        return gg[0] + JJ[0].T @ lam[0]
    gfd, Jfd, Hfd = np.empty(ipm.nw), np.empty((ipm.m, ipm.nw)), np.empty((ipm.nw, ipm.nw))
    for i in range(ipm.nw):
        e = np.zeros((1, ipm.nw))
        e[0, i] = h * max(1., abs(w[0, i]))
        fp, cp = ipm.eval_fc(w + e, data)
        fm, cm = ipm.eval_fc(w - e, data)
        gfd[i] = (fp[0] - fm[0]) / (2 * e[0, i])
        Jfd[:, i] = (cp[0] - cm[0]) / (2 * e[0, i])
        Hfd[:, i] = (lag_grad(w + e) - lag_grad(w - e)) / (2 * e[0, i])
    np.testing.assert_allclose(g[0], gfd, rtol=1e-6, atol=1e-6 * np.abs(g).max())
    np.testing.assert_allclose(J[0], Jfd, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(W[0], Hfd, rtol=1e-5, atol=1e-6 * np.abs(W).max())
    assert np.abs(W[0] - W[0].T).max() < 1e-12 * np.abs(W).max()


def test_scaling_leaves_the_estimate_alone():
    """Costs act on un-scaled quantities (modeling.py:665-672) and the bounds are divided by the scaling (mhe.py:640-655): the
    scaled problem is the un-scaled one in other units.  `u_meas` enters the scaled model un-divided (mhe.py:352 vs :242), so the
    equivalent call hands over u / su.  The row adds the SCALED noise variable to the SCALED state (mhe.py:733-740) while the cost
    sees w * w_scaling: the two only describe the same noise when w_scaling = x_scaling, which is what this case uses."""
    N = 5
    xa, um, ym, _ = c3_data(2, N=N, seed=3)
    a = MheIpm(oracle_mhe(dict(C3B, N=N)), IpmOptions(tol=1e-11)).solve(xa, C3B['p'], um, ym)
    spec = dict(C3B, N=N, x_scaling=[.5, 20., .1, .2], w_scaling=[.5, 20., .1, .2], u_scaling=[.1, .05])
    pb = oracle_mhe(spec)
    b = MheIpm(pb, IpmOptions(tol=1e-11)).solve(xa, C3B['p'], um / pb.su, ym)
    assert np.all(a['status'] == 1) and np.all(b['status'] == 1)
    np.testing.assert_allclose(b['f'], a['f'], rtol=1e-6)          # the bound relaxation (1e-8 max(1, |b|)) is not scale-invariant
    np.testing.assert_allclose(b['X'] * pb.sx, a['X'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(b['Wn'] * pb.sw, a['Wn'], atol=2e-7)
    np.testing.assert_allclose(b['x_opt'], a['x_opt'], rtol=1e-5, atol=1e-6)
