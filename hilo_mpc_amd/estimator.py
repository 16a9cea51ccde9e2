"""Batched Kalman / extended / unscented Kalman filters on the GPU.

API mirror of `hilo_mpc/modules/estimator/kf.py` (+ `_Estimator`, estimator/base.py) with a leading batch axis:
`setup()`, `Q`/`R` setters (scalar / vector -> diagonal, base.py:105-125), `set_initial_guess(x0, P0)` (P0
defaults to the identity, base.py:133-134), `set_initial_parameter_values(p)`, `estimate(y=, u=, p=)` (predict
then update, kf.py:258-265), and the pass-throughs `predict(xP, up, Q)` / `update(pred, y, up, R)` on packed
`[x|P]` tiles (kf.py:309-325).  All arithmetic runs in libhilo_hip.so (hilo_kf_*); nothing is computed on the host.
"""
import ctypes as C
import os
import warnings

import numpy as np
import torch

from . import _lib
from ._device import device, to_dev, ptr, stream_ptr, like_input

KIND = {'Kalman filter': 0, 'extended Kalman filter': 1, 'unscented Kalman filter': 2}


class _Solution:
    """Minimal stand-in for the reference's TimeSeries key language used with filters
    (`get_by_id('x:f')`, `['x:f']`, base.py:2206-2211,2426-2470): last state, covariance and output."""

    def __init__(self):
        self._d = {}

    def _set(self, **kw):
        self._d.update(kw)

    def _set_tile(self, out, yp):
        """Several steps per call, device results: the packed tiles [steps, B, n_x, n_x + 1] = [x | P] and the predicted outputs; the
        views x / P are made when asked for (a call's host time is counted in microseconds)."""
        self._d = {'x': (out, 0), 'P': (out, 1), 'y': yp}

    def get_by_id(self, key):
        name = key.split(':')[0]
        if name not in self._d:
            raise KeyError(key)
        v = self._d[name]
        if isinstance(v, tuple):
            v = v[0].select(3, 0) if v[1] == 0 else v[0].narrow(3, 1, v[0].shape[3] - 1)
        # device results are views of the filter's resident ping-pong tiles: what the caller gets is a copy of its own, taken
        # when it asks (a value collected per step must not change two steps later)
        return v.clone() if isinstance(v, torch.Tensor) else v

    __getitem__ = get_by_id


def _cov(v, n, dev):
    """base.py:105-125."""
    if isinstance(v, torch.Tensor):
        t = v.to(device=dev, dtype=torch.float64)
        if t.ndim >= 2 and t.shape[-1] == n and t.shape[-2] == n:
            return t.contiguous()
        v = t.cpu().numpy()
    a = np.asarray(v, dtype=float)
    if a.ndim == 0:
        a = np.eye(n) * float(a)
    elif a.ndim == 1:
        a = np.eye(n) * float(a[0]) if (a.size == 1 and n > 1) else np.diag(a)
    if a.shape[-2:] != (n, n):
        raise ValueError(f"Dimension mismatch. Supplied dimension is {a.shape[-2]}x{a.shape[-1]}, but required "
                         f"dimension is {n}x{n}.")
    return to_dev(a, dev)


class _KalmanFilter:
    _type = None

    def __init__(self, model, id=None, name=None, plot_backend=None, square_root_form=True, device_index=None,
                 n_sub=None):
        if not model._is_setup:                                   # estimator/base.py:74-76
            raise RuntimeError(f"Model is not set up. Run Model.setup() before passing it to the "
                               f"{'Kalman filter' if 'Kalman' in (self._type or '') else self._type}.")
        self._model = model
        self.name = name
        self._alpha, self._beta, self._kappa = 1e-3, 2., 0.
        self._handle = None
        self._dev_index = device_index
        self._n_sub = n_sub
        self._Q = self._R = None
        self._x = self._P = self._p = None
        self.solution = _Solution()

    type = property(lambda self: self._type)

    # ---- set-up ------------------------------------------------------------------------------
    def setup(self, **kwargs):
        m = self._model
        if not m._is_setup:
            m.setup()
        # HILO_JIT_COMPILE_ONLY=1 (image builds without a GPU): a filter on a model written as expressions is compiled into the
        # cache and setup() stops; other filters have nothing to compile
        compile_only = bool(os.environ.get('HILO_JIT_COMPILE_ONLY'))
        self._dev = None if compile_only else device(self._dev_index)
        desc = _lib.KfDesc()
        desc.model_id = m.model_id
        desc.kind = KIND[self._type]
        desc.continuous = 0 if m.discrete else 1          # kf.py:95-98
        desc.erk_order = m.erk_order if m.erk_order else 4
        desc.n_sub = self._n_sub if self._n_sub else (m.n_sub if m.discrete else 8)
        if m.name == 'lti':
            desc.lti_nx, desc.lti_nu, desc.lti_ny = m.n_x, m.n_u, m.n_y
        desc.dt = m.dt
        desc.alpha, desc.beta, desc.kappa = self._alpha, self._beta, self._kappa
        if getattr(m, '_symbolic', False):
            # a model written as expressions: its functor is compiled at setup (csrc/hilo_jit.hip) like the controllers' problems
            if not m.n_y:
                raise RuntimeError("The model has no measurement equations (set_measurement_equations)")
            self._user_source = m.user_source()
            desc.user_source = self._user_source.encode()
            # learned terms (Model.substitute_from): the trained GPs behind hilo_user_gp[k] of the source, like the controllers'
            gps = list(getattr(m, '_gps', None) or [])
            if len(gps) > 4:
                raise NotImplementedError(f"a filter model with {len(gps)} learned terms: at most 4 are offloaded (hilo_kf_desc.user_gp)")
            desc.n_user_gp = len(gps)
            for k, g in enumerate(gps):
                # (compile-only mode never dereferences the handles: any non-NULL value)
                desc.user_gp[k] = 1 if compile_only else (g._handle.value if hasattr(g._handle, 'value') else g._handle)
            self._gps = gps           # keeps the GP objects alive until the handle is created
        h = C.c_void_p()
        if compile_only:
            if desc.user_source:
                rc = _lib.lib().hilo_kf_create(C.byref(desc), 0, C.byref(h))
                if rc != _lib.COMPILED_ONLY:
                    _lib.check(rc)
            return
        _lib.check(_lib.lib().hilo_kf_create(C.byref(desc), self._dev.index, C.byref(h)))
        if self._handle is not None:
            _lib.lib().hilo_kf_destroy(self._handle)
        self._handle = h
        self._n_x, self._n_u, self._n_p, self._n_y = m.n_x, m.n_u, m.n_p, m.n_y
        self._pred_w = (2 + 3 * m.n_x) if self._type == 'unscented Kalman filter' else m.n_x + 1
        self._Q = torch.zeros(m.n_x, m.n_x, dtype=torch.float64, device=self._dev)   # kf.py:276-277
        self._R = torch.zeros(m.n_y, m.n_y, dtype=torch.float64, device=self._dev)
        if m.name == 'lti':
            self._p = to_dev(m.lti_parameters(), self._dev, (1, -1))

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().hilo_kf_destroy(self._handle)
        except Exception:
            pass

    def _check_setup(self):
        if self._handle is None:
            raise RuntimeError(f"{self._type[0].upper() + self._type[1:]} is not set up. Run "
                               f"{self.__class__.__name__}.setup() before running simulations.")

    # ---- tuning ------------------------------------------------------------------------------
    @property
    def Q(self):
        return self._Q

    @Q.setter
    def Q(self, v):
        self._check_setup()
        self._Q = _cov(v, self._n_x, self._dev)

    @property
    def R(self):
        return self._R

    @R.setter
    def R(self, v):
        self._check_setup()
        self._R = _cov(v, self._n_y, self._dev)

    def set_initial_guess(self, x0, P0=None):
        """base.py:127-142.  x0: [n_x] or [B, n_x]; P0: scalar / vector / matrix (per batch allowed)."""
        self._check_setup()
        x = to_dev(x0, self._dev)
        x = x.reshape(1, -1) if x.ndim <= 1 else x
        if x.shape[1] != self._n_x:
            raise ValueError(f"Dimension mismatch. Supplied dimension is {x.shape[1]}, but required dimension is "
                             f"{self._n_x}.")
        B = x.shape[0]
        P = _cov(1. if P0 is None else P0, self._n_x, self._dev)
        self._x = x.contiguous()
        self._P = P.expand(B, -1, -1).contiguous() if P.ndim == 2 else P.contiguous()

    def set_initial_parameter_values(self, p):
        self._check_setup()
        p = to_dev(p, self._dev)
        self._p = p.reshape(1, -1) if p.ndim <= 1 else p

    # ---- packed-tile pass-throughs (kf.py:309-325) ------------------------------------------------
    def _tile(self, t, width):
        t = to_dev(t, self._dev)
        if t.ndim == 2:
            t = t[None]
        if t.shape[1:] != (self._n_x, width):
            raise ValueError(f"expected a packed tile of shape [B, {self._n_x}, {width}], got {tuple(t.shape)}")
        return t.contiguous()

    def _up(self, up, B):
        n = self._n_u + self._n_p
        if n == 0:
            return None, 0
        t = to_dev(up, self._dev)
        t = t.reshape(1, -1) if t.ndim <= 1 else t
        if t.shape[1] != n:
            raise ValueError(f"Dimension mismatch in [u; p]: got {t.shape[1]}, the model has {self._n_u} inputs and "
                             f"{self._n_p} parameters")
        if t.shape[0] not in (1, B):
            raise ValueError(f"[u; p] has batch {t.shape[0]}, expected 1 or {B}")
        return t.contiguous(), (0 if t.shape[0] == 1 else n)

    @staticmethod
    def _cov_stride(M, B):
        if M.ndim == 2:
            return 0
        if M.shape[0] not in (1, B):
            raise ValueError(f"covariance batch {M.shape[0]} does not match {B}")
        return 0 if M.shape[0] == 1 else M.shape[1] * M.shape[2]

    def predict(self, xP, up=None, Q=None):
        """`prediction_step(x0, p, Q)` (kf.py:129-133; UKF :550-554 returns [x|P|X])."""
        self._check_setup()
        t = self._tile(xP, self._n_x + 1)
        B = t.shape[0]
        upt, us = self._up(up, B)
        Qt = self._Q if Q is None else _cov(Q, self._n_x, self._dev)
        out = torch.empty(B, self._n_x, self._pred_w, dtype=torch.float64, device=self._dev)
        _lib.check(_lib.lib().hilo_kf_predict(self._handle, B, ptr(t), ptr(upt), us, ptr(Qt),
                                              self._cov_stride(Qt, B), ptr(out), stream_ptr(self._dev)))
        return like_input(out if np.ndim(xP) == 3 else out[0], xP)

    def update(self, pred, y, up=None, R=None):
        """`update_step(x0, y, p, R)` (kf.py:182-186; UKF :600-604).  Returns ([x+|P+], y_pred)."""
        self._check_setup()
        t = self._tile(pred, self._pred_w)
        B = t.shape[0]
        upt, us = self._up(up, B)
        Rt = self._R if R is None else _cov(R, self._n_y, self._dev)
        yt = to_dev(y, self._dev).reshape(-1, self._n_y)
        if yt.shape[0] != B:
            raise ValueError(f"Dimension mismatch. Supplied {yt.shape[0]} measurement vectors for a batch of {B}.")
        out = torch.empty(B, self._n_x, self._n_x + 1, dtype=torch.float64, device=self._dev)
        yp = torch.empty(B, self._n_y, dtype=torch.float64, device=self._dev)
        _lib.check(_lib.lib().hilo_kf_update(self._handle, B, ptr(t), ptr(yt.contiguous()), ptr(upt), us, ptr(Rt),
                                             self._cov_stride(Rt, B), ptr(out), ptr(yp), stream_ptr(self._dev)))
        if np.ndim(pred) == 3:
            return like_input(out, pred), like_input(yp, pred)
        return like_input(out[0], pred), like_input(yp[0].reshape(-1, 1), pred)

    # ---- several steps per call: `self._function.mapaccum(steps)` (kf.py:296-306) -------------------
    def _estimate_steps(self, y, u, p, steps):
        """y [steps, B, n_y]; u [steps, B, n_u] or [B, n_u] (held over the steps); p like a single step.  One launch for all
        steps (hilo_kf_steps); the solution holds the sequences x [steps, B, n_x], P [steps, B, n_x, n_x], y [steps, B, n_y]."""
        dev = self._dev
        yt = to_dev(y, dev)
        B = self._x.shape[0]
        if yt.numel() % (steps * self._n_y):
            raise ValueError(f"Dimension mismatch for variable y. Supplied dimension is {yt.numel()}, but required dimension "
                             f"is a multiple of {steps * self._n_y}.")
        yt = yt.reshape(steps, -1, self._n_y).contiguous()
        if yt.shape[1] != B:
            if B != 1:
                raise ValueError(f"Dimension mismatch. Supplied {yt.shape[1]} measurement vectors per step for {B} filters.")
            B = yt.shape[1]
            self._x = self._x.expand(B, -1).contiguous()
            self._P = self._P.expand(B, -1, -1).contiguous()
        # inputs and parameters go over as they are - separate arrays like the separate arguments of the reference's function
        # (kf.py:130; hilo_kf_steps_split): rows per instance or ONE row shared by the batch (stride 0), the inputs per step or held
        ut = pt = None
        us = ustep = ps = 0
        if self._n_p:
            pt = self._p if p is None else to_dev(p, dev)
            if pt is None:
                raise RuntimeError("No parameter values supplied. Please run set_initial_parameter_values() or pass p=.")
            pt = pt.reshape(-1, self._n_p)
            if pt.shape[0] not in (1, B):
                raise ValueError(f"Dimension mismatch for variable p: {pt.shape[0]} parameter vectors for {B} filters.")
            ps = self._n_p if pt.shape[0] == B and B > 1 else 0
        if self._n_u:
            if u is None:
                raise RuntimeError("No input data supplied.")
            ut = to_dev(u, dev)
            if ut.numel() == steps * B * self._n_u and steps > 1:
                us, ustep = self._n_u, B * self._n_u
            else:
                ut = ut.reshape(-1, self._n_u)
                if ut.shape[0] not in (1, B):
                    raise ValueError(f"Dimension mismatch for variable u: {ut.shape[0]} input vectors for {B} filters.")
                us = self._n_u if ut.shape[0] == B and B > 1 else 0
        xP = self._packed_tile(B)
        out = torch.empty(steps, B, self._n_x, self._n_x + 1, dtype=torch.float64, device=dev)
        yp = torch.empty(steps, B, self._n_y, dtype=torch.float64, device=dev)
        _lib.check(_lib.lib().hilo_kf_steps_split(self._handle, B, int(steps), ptr(xP), ptr(yt), ptr(ut), us, ustep, ptr(pt), ps,
                                                  ptr(self._Q), self._cov_stride(self._Q, B), ptr(self._R), self._cov_stride(self._R, B),
                                                  ptr(out), 1, ptr(yp), stream_ptr(dev)))
        st = self._state_tile = out.select(0, steps - 1)        # the last step's packed tile is the filter state: x and P are views of it
        self._x, self._P = st.select(2, 0), st.narrow(2, 1, self._n_x)
        if isinstance(y, torch.Tensor):
            self.solution._set_tile(out, yp)
        else:
            self.solution._set(x=out[:, :, :, 0].cpu().numpy(), P=out[:, :, :, 1:].cpu().numpy(), y=yp.cpu().numpy())
        return self.solution

    def _packed_tile(self, B):
        """The filter state as a packed [B, nx, nx+1] tile [x | P]: the tile the last step wrote when x and P still are its views
        (no copy), otherwise (set_initial_guess, assignment from outside) packed once."""
        t = getattr(self, '_state_tile', None)
        if (t is not None and t.shape[0] == B and self._x.data_ptr() == t.data_ptr() and self._P.data_ptr() == t.data_ptr() + 8
                and self._x.stride() == (t.stride(0), t.stride(1)) and self._P.stride() == t.stride()):
            return t
        t = torch.empty(B, self._n_x, self._n_x + 1, dtype=torch.float64, device=self._dev)
        t[:, :, 0] = self._x
        t[:, :, 1:] = self._P
        return t

    # ---- estimate (kf.py:279-307) ----------------------------------------------------------------
    def estimate(self, y=None, u=None, p=None, **kwargs):
        self._check_setup()
        if self._x is None:
            raise RuntimeError("No initial guess for the states found. Please set initial guess before running "
                               "the Kalman filter!")
        if y is None:
            raise RuntimeError("No measurement data supplied.")
        if int(kwargs.get('steps', 1) or 1) > 1:
            # (`inputs_unchanged=` is accepted and has nothing left to do here: inputs and parameters are handed over where they are)
            return self._estimate_steps(y, u, p, int(kwargs['steps']))
        B = self._x.shape[0]
        for name, val, n in (('y', y, self._n_y), ('u', u, self._n_u), ('p', p, self._n_p)):     # base.py `_process_inputs`
            if val is not None and n:
                size = val.numel() if isinstance(val, torch.Tensor) else int(np.size(val))
                if size % n:
                    raise ValueError(f"Dimension mismatch for variable {name}. Supplied dimension is {size}, but required "
                                     f"dimension is {n}.")
        yt = to_dev(y, self._dev).reshape(-1, self._n_y)
        if yt.shape[0] != B:
            if B == 1:                                   # first call decides the batch size
                B = yt.shape[0]
                self._x = self._x.expand(B, -1).contiguous()
                self._P = self._P.expand(B, -1, -1).contiguous()
            else:
                raise ValueError(f"Dimension mismatch. Supplied {yt.shape[0]} measurement vectors for {B} filters.")
        if p is not None:
            pt = to_dev(p, self._dev)
            pt = pt.reshape(1, -1) if pt.ndim <= 1 else pt
        else:
            pt = self._p
        if self._n_p and pt is None:
            raise RuntimeError("No parameter values supplied. Please run set_initial_parameter_values() or pass p=.")
        ut = None
        if self._n_u:
            if u is None:
                raise RuntimeError("No input data supplied.")
            ut = to_dev(u, self._dev).reshape(-1, self._n_u)
        # [u; p] of kf.py:130 in a buffer the filter keeps: per step only the rows that changed are written (no allocation,
        # no concatenation kernel); the packed [x|P] tile ping-pongs between two resident buffers, x and P are views of it
        nup = self._n_u + self._n_p
        upt, us = None, 0
        unchanged = bool(kwargs.get('inputs_unchanged', False))
        if nup:
            buf = getattr(self, '_up_buf', None)
            if buf is None or buf.shape[0] != B:
                buf = self._up_buf = torch.empty(B, nup, dtype=torch.float64, device=self._dev)
                self._up_p_src = self._up_u_src = None
            if self._n_u:
                ukey = (id(u), u._version) if isinstance(u, torch.Tensor) and u.device == buf.device else None
                if not unchanged or ukey is None or getattr(self, '_up_u_src', None) != ukey:
                    buf[:, :self._n_u] = ut
                    self._up_u_src, self._up_u_ref = ukey, u
            if self._n_p:
                # estimate(..., inputs_unchanged=True): device tensors handed over again (same live object, same version counter)
                # are not copied again - the caller states that they hold the same values (a write through a raw pointer does not bump the
                # version counter); host data always is
                src = self._p if p is None else p
                key = (id(src), src._version) if isinstance(src, torch.Tensor) and src.device == buf.device else None
                if not unchanged or key is None or self._up_p_src != key:
                    buf[:, self._n_u:] = pt
                    self._up_p_src, self._up_p_ref = key, src          # (kept alive: an id is only unique among live objects)
            upt, us = buf, nup
        # the packed [x|P] tile ping-pongs between two resident buffers, x and P are views of the one written last
        xP = self._packed_tile(B)
        tiles = getattr(self, '_xP_bufs', None)
        if tiles is None or tiles[0].shape[0] != B:
            tiles = self._xP_bufs = [torch.empty(B, self._n_x, self._n_x + 1, dtype=torch.float64, device=self._dev) for _ in range(2)]
        out = tiles[1] if xP.data_ptr() == tiles[0].data_ptr() else tiles[0]
        yp = torch.empty(B, self._n_y, dtype=torch.float64, device=self._dev)
        _lib.check(_lib.lib().hilo_kf_step(self._handle, B, ptr(xP), ptr(yt.contiguous()), ptr(upt), us,
                                           ptr(self._Q), self._cov_stride(self._Q, B), ptr(self._R),
                                           self._cov_stride(self._R, B), ptr(out), ptr(yp),
                                           stream_ptr(self._dev)))
        self._state_tile = out
        self._x = out[:, :, 0]          # strided views of the resident tile (zero copy)
        self._P = out[:, :, 1:]
        host = not isinstance(y, torch.Tensor)
        cv = (lambda t: t.cpu().numpy()) if host else (lambda t: t)
        xs = cv(self._x)
        self.solution._set(x=xs.T if (host and B == 1) else xs, P=cv(self._P), y=cv(yp))
        return self.solution

    # accessors of the reference's estimator base (estimator/base.py:60-200)
    n_x = property(lambda s: s._model.n_x)
    n_u = property(lambda s: s._model.n_u)
    n_p = property(lambda s: s._model.n_p)
    n_y = property(lambda s: s._model.n_y)
    n_z = property(lambda s: getattr(s._model, 'n_z', 0))
    n_p_est = property(lambda s: 0)                       # the filters estimate states only
    process_noise_covariance = property(lambda s: s._Q)
    measurement_noise_covariance = property(lambda s: s._R)
    error_covariance = property(lambda s: None if s._P is None else s._P.clone())

    def is_setup(self):
        return self._handle is not None

    # filter state on the device: copies (the resident tiles behind them are overwritten two steps later)
    @property
    def x(self):
        return None if self._x is None else self._x.clone()

    @property
    def P(self):
        return None if self._P is None else self._P.clone()


class KalmanFilter(_KalmanFilter):
    """kf.py:328-367."""
    _type = 'Kalman filter'

    def __init__(self, model, **kw):
        if not model.is_linear():
            raise ValueError("The supplied model is nonlinear. Please use an estimator targeted at the estimation of "
                             "nonlinear systems.")
        super().__init__(model, **kw)


class ExtendedKalmanFilter(_KalmanFilter):
    """kf.py:370-410."""
    _type = 'extended Kalman filter'

    def __init__(self, model, **kw):
        if model.is_linear():
            warnings.warn("The supplied model is linear. For better efficiency use an observer targeted at the "
                          "estimation of linear systems.")
        super().__init__(model, **kw)


class UnscentedKalmanFilter(_KalmanFilter):
    """kf.py:413-640."""
    _type = 'unscented Kalman filter'

    def __init__(self, model, alpha=None, beta=None, kappa=None, **kw):
        if model.is_linear():
            warnings.warn("The supplied model is linear. For better efficiency use an observer targeted at the "
                          "estimation of linear systems.")
        super().__init__(model, **kw)
        self.alpha = .001 if alpha is None else alpha
        self.beta = 2. if beta is None else beta
        self.kappa = 0. if kappa is None else kappa

    alpha = property(lambda s: float(s._alpha))
    beta = property(lambda s: float(s._beta))
    kappa = property(lambda s: float(s._kappa))

    @alpha.setter
    def alpha(self, v):
        if v <= 0. or v > 1.:
            raise ValueError(f"The parameter alpha needs to lie in the interval (0, 1]. Supplied alpha is {float(v)}.")
        self._alpha = v

    @beta.setter
    def beta(self, v):
        self._beta = v

    @kappa.setter
    def kappa(self, v):
        if v < 0:
            raise ValueError(f"The parameter kappa needs to be greater or equal to 0. Supplied kappa is {float(v)}.")
        self._kappa = v
