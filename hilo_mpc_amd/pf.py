"""Particle filter: `ParticleFilter` of the reference (hilo_mpc/modules/estimator/pf.py) for a batch of filters.

The reference builds ONE function at `setup()` - propagate the particles through the model, add the noise samples, evaluate the
measurement map and the normalised likelihood of every particle (pf.py:103-166, :300-318) - and calls it once per `estimate()`
(:372); everything random (noise samples, resampling, roughening) is drawn with numpy on the host (:364-368, :404-415).  The same
split here: the function, the resampling gather and the statistics of the particle set run on the device
(`hilo_pf_function / hilo_pf_resample / hilo_pf_stats`, csrc/hilo_kf_kernel.h), the random numbers come from numpy's global
generator in the reference's order - process noise from the sampling function (`pdf`, default `lhsnorm`), measurement noise
`sqrt(R) @ randn`, the uniforms `np.random.choice` draws for the resampling, the roughening noise - so a seeded run consumes the
stream exactly like the reference (for a batch: filter after filter inside each of these stages).
"""
import warnings

import numpy as np
import torch

from . import _lib
from ._device import ptr, stream_ptr, to_dev
from .estimator import KIND, _KalmanFilter

KIND['particle filter'] = 1          # the handle only carries the model, the sampling interval and the discretisation


def lhsnorm(mu, sigma, n):
    """Latin-hypercube sample of a normal distribution, [n, dim] (pf.py:425-447): a correlated normal sample only supplies the
    RANKS per component; each component then takes one point per probability stratum, (rank - U(0, 1)) / n, mapped through the
    inverse normal cdf with the marginal mean and variance `diag(sigma)`.  Draws from numpy's global generator in the reference's
    order: `multivariate_normal(mu, sigma, size=n)`, then `rand(n, dim)`."""
    from scipy.stats import norm
    mu = np.asarray(mu, dtype=float)
    sigma = np.asarray(sigma, dtype=float)
    z = np.random.multivariate_normal(mu, sigma, size=n)
    ranks = np.empty_like(z)
    np.put_along_axis(ranks, np.argsort(z, axis=0), np.arange(1., n + 1.)[:, None], axis=0)
    strata = (ranks - np.random.rand(n, mu.size)) / n
    return norm.ppf(strata, loc=mu[None, :], scale=np.sqrt(np.diag(sigma))[None, :])


class ParticleFilter(_KalmanFilter):
    """Particle filter (PF) class for state estimation (pf.py:36-91)."""
    _type = 'particle filter'

    def __init__(self, model, id=None, name=None, plot_backend=None, variant=None, roughening=False, prior_editing=False,
                 device_index=None, **kwargs):
        if model.is_linear():
            warnings.warn("The supplied model is linear. For better efficiency use an observer targeted at the "
                          "estimation of linear systems.")
        super().__init__(model, id=id, name=name, plot_backend=plot_backend, device_index=device_index)
        self._variant = variant
        self._roughening = roughening
        self._prior_editing = prior_editing
        if self._roughening or self._prior_editing:
            K = kwargs.get('K')
            if K is None:
                K = .2
            self._roughening_tuning_param = K
        self._sample_size = 15
        self._pdf = lhsnorm
        self._transpose_pdf = None
        self._nx_model = model.n_x
        self._X = None

    # ---- sampling function (pf.py:191-248) -----------------------------------------------------------------------------------
    @property
    def probability_density_function(self):
        return self._pdf

    # what the reference expects of a sampler `X = pdf(mean, covariance, n)` when it is annotated (pf.py:197-249): position -> type
    # and the role named in the message
    _SAMPLER_SIGNATURE = ((np.ndarray, "The 1st argument to the probability density function (pdf) needs to be the 'mean'"),
                          (np.ndarray, "The 2nd argument to the probability density function (pdf) needs to be the 'covariance'"),
                          (int, "The 3rd argument to the probability density function (pdf) needs to be the 'sample size'"),
                          (np.ndarray, "The return value of the probability density function (pdf) needs to be a 'random sample'"))

    def _sampler_orientation(self, pdf):
        """Draw once from N(0, I): does the sampler return particles as columns (False) or as rows (True)?"""
        n, size = self._nx_model, self._sample_size
        try:
            shape = tuple(np.shape(pdf(np.zeros(n), np.eye(n), size)))
            if shape == (n, size):
                return False
            if shape == (size, n):
                return True
            raise ValueError(f"Dimension mismatch. Expected dimension {n}x{size}, got {shape[0]}x{shape[1]}.")
        except Exception as err:
            raise RuntimeError(f"The following exception was raised\n"
                               f"   {type(err).__name__}: '{err.args[0]}'.\nPlease make sure that the "
                               f"supplied probability density function (pdf) has the following arguments\n"
                               f"   mu - mean of the pdf (type: numpy.ndarray),\n"
                               f"   sigma - covariance of the mean (type: numpy.ndarray),\n"
                               f"   n - sample size (type: int),\n"
                               f"and the following return value\n"
                               f"   X - random sample (type: numpy.ndarray).")

    @probability_density_function.setter
    def probability_density_function(self, pdf):
        """A fully annotated sampler (three arguments and the return value) is checked by its annotations and trusted to return
        particles as columns; anything else is probed with one draw (same decisions and messages as pf.py:197-249)."""
        if not callable(pdf):
            raise ValueError(f"Probability density function of the {self.type} needs to be callable.")
        notes = dict(getattr(pdf, '__annotations__', None) or {})
        if len(notes) == len(self._SAMPLER_SIGNATURE) and 'return' in notes:
            for found, (want, what) in zip(notes.values(), self._SAMPLER_SIGNATURE):
                if found is not want:
                    raise TypeError(f"{what} with type {want.__name__}.")
        else:
            self._transpose_pdf = self._sampler_orientation(pdf)
        self._pdf = pdf

    pdf = probability_density_function

    @property
    def variant(self):
        return self._variant

    @variant.setter
    def variant(self, variant):
        self._variant = variant

    @property
    def sample_size(self):
        return self._sample_size

    @sample_size.setter
    def sample_size(self, sample_size):
        self._sample_size = sample_size

    n_samples = sample_size

    # ---- setup (pf.py:279-338) -----------------------------------------------------------------------------------------------
    def setup(self, **kwargs):
        n_s = kwargs.get('n_samples')
        if n_s is not None:
            self._sample_size = n_s
        if not 2 <= int(self._sample_size) <= 8192:
            raise ValueError("the particle filter is built for 2 to 8192 particles per filter")
        super().setup()
        m = self._model
        if m.n_y == 0:
            warnings.warn(f"The model has no measurement equations, I am assuming measurements of all states "
                          f"{m.dynamical_state_names} are available.")
        self._n_ye = m.n_y if m.n_y else m.n_x
        self._R = torch.zeros(self._n_ye, self._n_ye, dtype=torch.float64, device=self._dev)     # pf.py:337-338
        self._X = None

    @_KalmanFilter.R.setter
    def R(self, v):
        from .estimator import _cov
        self._check_setup()
        self._R = _cov(v, self._n_ye, self._dev)

    # ---- the reference's function on the device ---------------------------------------------------------------------------------
    def _sample(self, mean, cov, n):
        """One draw of the sampling function as [n, dim] (particle-major).  Orientation like the reference (pf.py:173-186,
        :365-368): a function set through the property was probed there; the default one is probed on first use - an
        (dim x n) result is taken as is, an (n x dim) one is transposed (the reference hard-codes dim = 2 in this probe)."""
        S = np.asarray(self._pdf(mean, cov, n))
        if self._transpose_pdf is None:
            if S.shape == (mean.size, n):
                self._transpose_pdf = False
            elif S.shape == (n, mean.size):
                self._transpose_pdf = True
            else:
                raise ValueError(f"Dimension mismatch. Expected dimension {mean.size}x{n}, got {S.shape[1]}x{S.shape[0]}.")
        return S if self._transpose_pdf else S.T

    def function(self, X, y, up, w, v, R=None):
        """`self._function(X=, y=, p=up, w=, v=, R=)` (pf.py:312-318) for a batch: particle arrays [B, N, n] (particle-major),
        y [B, n_y], up [B or 1, n_u + n_p].  Returns (X_prop, Y, q) on the device."""
        self._check_setup()
        N, nx, ny = int(self._sample_size), self._n_x, self._n_ye
        Xt = to_dev(X, self._dev).reshape(-1, N, nx).contiguous()
        B = Xt.shape[0]
        yt = to_dev(y, self._dev).reshape(B, ny).contiguous()
        wt = to_dev(w, self._dev).reshape(B, N, nx).contiguous()
        vt = to_dev(v, self._dev).reshape(B, N, ny).contiguous()
        upt, us = self._up(up, B)
        from .estimator import _cov
        Rt = self._R if R is None else _cov(R, ny, self._dev)
        Xp, Y = torch.empty_like(Xt), torch.empty(B, N, ny, dtype=torch.float64, device=self._dev)
        q = torch.empty(B, N, dtype=torch.float64, device=self._dev)
        _lib.check(_lib.lib().hilo_pf_function(self._handle, B, N, ptr(Xt), ptr(yt), ptr(upt), us, ptr(wt), ptr(vt), ptr(Rt),
                                               self._cov_stride(Rt, B), ptr(Xp), ptr(Y), ptr(q), stream_ptr(self._dev)))
        return Xp, Y, q

    def _resample(self, Xp, Y, q, uni):
        B, N, _ = Xp.shape
        Xr, Yr = torch.empty_like(Xp), torch.empty_like(Y)
        ind = torch.empty(B, N, dtype=torch.int32, device=Xp.device)
        _lib.check(_lib.lib().hilo_pf_resample(self._handle, B, N, ptr(Xp), ptr(Y), ptr(q), ptr(uni.contiguous()), ptr(Xr), ptr(Yr),
                                               ptr(ind), stream_ptr(self._dev)))
        return Xr, Yr, ind

    def _stats(self, X, Y, add=None):
        B, N, nx = X.shape
        dev = self._dev
        xm, ym = torch.empty(B, nx, dtype=torch.float64, device=dev), torch.empty(B, self._n_ye, dtype=torch.float64, device=dev)
        P = torch.empty(B, nx, nx, dtype=torch.float64, device=dev)
        lo, hi = torch.empty(B, nx, dtype=torch.float64, device=dev), torch.empty(B, nx, dtype=torch.float64, device=dev)
        _lib.check(_lib.lib().hilo_pf_stats(self._handle, B, N, ptr(X), ptr(Y), ptr(add), ptr(xm), ptr(ym), ptr(P), ptr(lo), ptr(hi),
                                            stream_ptr(dev)))
        return xm, ym, P, lo, hi

    # ---- estimate (pf.py:340-422) --------------------------------------------------------------------------------------------
    def estimate(self, y=None, u=None, p=None, **kwargs):
        self._check_setup()
        if self._x is None:
            raise RuntimeError("No initial guess for the states found. Please set initial guess before running "
                               "the particle filter!")
        if y is None:
            raise RuntimeError("No measurement data supplied.")
        N, nx, ny, dev = int(self._sample_size), self._n_x, self._n_ye, self._dev
        yt = to_dev(y, dev).reshape(-1, ny)
        B = yt.shape[0]
        if self._x.shape[0] not in (1, B):
            raise ValueError(f"Dimension mismatch. Supplied {B} measurement vectors for {self._x.shape[0]} filters.")
        if self._X is None or self._X.shape[0] != B:                 # `_initial_sample` (pf.py:168-188)
            x0 = self._x.expand(B, -1).cpu().numpy()
            P0 = self._P.expand(B, -1, -1).cpu().numpy()
            self._X = to_dev(np.stack([self._sample(x0[b], P0[b], N) for b in range(B)]), dev)
        pt = self._p if p is None else to_dev(p, dev)
        if self._n_p and pt is None:
            raise RuntimeError("No parameter values supplied. Please run set_initial_parameter_values() or pass p=.")
        up = None
        if self._n_u + self._n_p:
            parts = []
            if self._n_u:
                if u is None:
                    raise RuntimeError("No input data supplied.")
                ut = to_dev(u, dev).reshape(-1, self._n_u)
                parts.append(ut.expand(B, -1) if ut.shape[0] == 1 else ut)
            if self._n_p:
                pt = pt.reshape(-1, self._n_p)
                parts.append(pt.expand(B, -1) if pt.shape[0] == 1 else pt)
            up = torch.cat(parts, dim=1).contiguous()
        Qh = self._Q.expand(B, -1, -1).cpu().numpy() if self._Q.ndim == 3 else np.broadcast_to(self._Q.cpu().numpy(), (B, nx, nx))
        Rh = self._R.expand(B, -1, -1).cpu().numpy() if self._R.ndim == 3 else np.broadcast_to(self._R.cpu().numpy(), (B, ny, ny))
        zx = np.zeros(nx)
        w = np.stack([self._sample(zx, Qh[b], N) for b in range(B)])                         # pf.py:365
        v = np.stack([(np.sqrt(Rh[b]) @ np.random.randn(ny, N)).T for b in range(B)])        # pf.py:367
        X = self._X
        Xp, Y, q = self.function(X, yt, up, w, v)
        if self._prior_editing:                                                              # pf.py:377-399
            sig6 = 6 * np.sqrt(np.stack([np.diag(Rh[b]) for b in range(B)]))                 # [B, ny]
            yh = yt.cpu().numpy()
            while True:
                need = np.any(np.abs(yh[:, None, :] - Y.cpu().numpy()) > sig6[:, None, :], axis=2)      # [B, N]
                if not need.any():
                    break
                _, _, _, lo, hi = self._stats(X, Y)
                spread = (hi - lo).cpu().numpy()
                add = np.zeros((B, N, nx))
                for b in range(B):
                    n_r = int(need[b].sum())
                    if n_r:
                        add[b, need[b]] = self._sample(zx, self._roughening_tuning_param * np.diag(spread[b]) * n_r ** (-1 / nx), n_r)
                X = X + to_dev(add, dev)
                Xp, Y, q = self.function(X, yt, up, w, v)
        # resampling (pf.py:404-407): the uniforms `np.random.choice` would draw, the search and the gather on the device
        uni = to_dev(np.stack([np.random.random_sample(N) for _ in range(B)]), dev)
        Xr, Yr, ind = self._resample(Xp, Y, q, uni)
        add = None
        if self._roughening:                                                                 # pf.py:409-415
            _, _, _, lo, hi = self._stats(Xr, Yr)
            spread = (hi - lo).cpu().numpy()
            add = to_dev(np.stack([self._sample(zx, self._roughening_tuning_param * np.diag(spread[b]) * N ** (-1 / nx), N)
                                   for b in range(B)]), dev)
        xm, ym, P, _, _ = self._stats(Xr, Yr, add)
        self._X, self._x, self._P = Xr, xm, P
        self._last = dict(X_prop=Xp, Y=Y, q=q, index=ind)
        host = not isinstance(y, torch.Tensor)
        cv = (lambda t: t.cpu().numpy()) if host else (lambda t: t)
        xs = cv(xm)
        # X like the reference's n_x x N matrix per filter
        self.solution._set(x=xs.T if (host and B == 1) else xs, X=cv(Xr.transpose(1, 2)), P=cv(P), y=cv(ym))
        return self.solution


__all__ = ['ParticleFilter', 'lhsnorm']
