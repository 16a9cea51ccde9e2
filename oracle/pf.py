"""Oracle: the particle filter of the reference, restated in numpy.

TEST INFRASTRUCTURE ONLY - never imported by the product package.   PARITY UNPINNED: tests/test_PFs.py checks the interface
(sampling-function setter, warnings, defaults), no filtered numbers - the filter draws from numpy's unseeded global generator.

Restated from hilo_mpc/modules/estimator/pf.py:
  * the function of `setup()` (:103-166, :300-318):  X_prop = Phi(X, u, p) + w,  Y = h(X_prop) + v,
    q = normpdf(Y; y, sqrt(R)) / sum (normpdf :99); one measurement in the reference - for several the joint likelihood of
    independent measurements (diagonal R) is used here and in the product;
  * `estimate` (:340-422): initial sample from the sampling function, process noise from the sampling function, measurement
    noise sqrt(R) @ randn, optional prior editing, `np.random.choice(N, N, p=q)`, optional roughening with
    K diag(max - min) N^(-1/n_x), mean / np.cov of the particle set;
  * `lhsnorm` (:425-447).
Random numbers come from numpy's GLOBAL generator in the reference's order, so that a seeded run is comparable draw by draw.
"""
import numpy as np
from scipy.stats import norm


def lhsnorm(mu, sigma, n):
    """pf.py:425-447, component by component: ranks of a correlated normal sample pick the stratum, a uniform draw the place
    inside it, the inverse normal cdf (marginal mean and standard deviation) the value."""
    dim = np.size(mu)
    sample = np.random.multivariate_normal(mu, sigma, size=n)        # only its ranks are used
    jitter = np.random.rand(n, dim)
    out = np.empty((n, dim))
    for k in range(dim):
        rank = np.empty(n)
        rank[np.argsort(sample[:, k])] = np.arange(1, n + 1)
        out[:, k] = norm.ppf((rank - jitter[:, k]) / n, loc=mu[k], scale=np.sqrt(sigma[k, k]))
    return out


def pf_function(model, dt, X, y, u, p, w, v, R):
    """X [N, nx] (particle-major), y [ny], w [N, nx], v [N, ny]; model: DISCRETE OracleModel.  Returns X_prop, Y, q."""
    N = X.shape[0]
    U = np.tile(np.asarray(u, dtype=float).reshape(1, -1), (N, 1))
    Pm = np.tile(np.asarray(p, dtype=float).reshape(1, -1), (N, 1))
    Xp = model.f(X, U, Pm, dt) + w
    Y = model.h(Xp, U, Pm, dt) + v
    sig = np.sqrt(np.diag(np.atleast_2d(R)))
    q = np.prod(np.exp(-.5 * ((Y - np.asarray(y, dtype=float).reshape(1, -1)) / sig) ** 2) / (np.sqrt(2 * np.pi) * sig), axis=1)
    return Xp, Y, q / q.sum()


class ParticleFilter:
    def __init__(self, model, dt, n_samples=15, roughening=False, prior_editing=False, K=.2, pdf=lhsnorm):
        self.model, self.dt, self.N = model, dt, n_samples
        self.roughening, self.prior_editing, self.K, self.pdf = roughening, prior_editing, K, pdf
        self.X = None
        self.Q = np.zeros((model.nx, model.nx))
        self.R = np.zeros((max(model.ny, 1),) * 2)

    def set_initial_guess(self, x0, P0):
        self.x0, self.P0 = np.asarray(x0, dtype=float), np.atleast_2d(np.asarray(P0, dtype=float))

    def estimate(self, y, u=(), p=()):
        nx, N = self.model.nx, self.N
        if self.X is None:
            self.X = self.pdf(self.x0, self.P0, N)
        w = self.pdf(np.zeros(nx), self.Q, N)
        v = (np.sqrt(self.R) @ np.random.randn(self.R.shape[0], N)).T
        X = self.X
        Xp, Y, q = pf_function(self.model, self.dt, X, y, u, p, w, v, self.R)
        if self.prior_editing:
            while True:
                need = np.any(np.abs(np.asarray(y).reshape(1, -1) - Y) > 6 * np.sqrt(np.diag(self.R)), axis=1)
                n_r = int(need.sum())
                if n_r == 0:
                    break
                dx = X.max(axis=0) - X.min(axis=0)
                X = X.copy()
                X[need] += self.pdf(np.zeros(nx), self.K * np.diag(dx) * n_r ** (-1 / nx), n_r)
                Xp, Y, q = pf_function(self.model, self.dt, X, y, u, p, w, v, self.R)
        ind = np.random.choice(N, size=N, replace=True, p=q)
        Xr, Yr = Xp[ind], Y[ind]
        if self.roughening:
            dx = Xr.max(axis=0) - Xr.min(axis=0)
            Xr = Xr + self.pdf(np.zeros(nx), self.K * np.diag(dx) * N ** (-1 / nx), N)
        self.X = Xr
        return dict(x=Xr.mean(axis=0), y=Yr.mean(axis=0), P=np.atleast_2d(np.cov(Xr.T)), X=Xr, X_prop=Xp, Y=Y, q=q, index=ind)
