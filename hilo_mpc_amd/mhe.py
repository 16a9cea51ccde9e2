"""Batched moving-horizon estimation on the GPU.

API mirror of `hilo_mpc.MHE` (`MovingHorizonEstimator`, hilo_mpc/modules/estimator/mhe.py) for the hot path:
`quad_arrival_cost.add_states(weights, guess)`, `quad_stage_cost.add_measurements(weights)`,
`quad_stage_cost.add_state_noise(weights)` (util/modeling.py:686-777), `horizon`, `set_box_constraints`,
`set_initial_guess`, `set_scaling`, `setup`, `add_measurements(y_meas, u_meas)` (mhe.py:274-309, ring buffer of the last
N samples), `estimate(x_arrival=None, p_arrival=None, v0=None)` (mhe.py:311-416; returns `(None, None)` until the
window is full, then `(x_N, p)` - the one-step-ahead state, mhe.py:381-384) - with a leading batch axis.

Scope: with or without state noise (`quad_stage_cost.add_state_noise`; without it the window's trajectory is a function of x_0 and
the parameters, mhe.py:599, :726-736); the reference's integration branches (mhe.py:512-593) - `'collocation'` (its default for a continuous
model: Radau / Legendre points of degree 1..4, collocation states in `v`, per-stage rows [collocation | continuity] in `lam_g`)
and `'discrete'` (pre-discretised model; `'rk4'` / `'erk'` discretise the model first); models of the device zoo
AND models written as expressions or text (`Model.set_dynamical_equations`, compiled with hiprtc at `setup()` around the
estimator's policy, csrc/hilo_mhe_policy.h); a subset of the measurements in the cost (`add_measurements(weights, names=)`,
modeling.py:686-712); model parameters either pinned by `p_lb == p_ub` or ESTIMATED (`quad_arrival_cost.add_parameters`,
bounds / guess / scaling of p, mhe.py:614-623; zoo models, models written as expressions, under collocation too) - `estimate` then
returns `(x_opt, p_opt)`.
"""
import ctypes as C
import warnings

import numpy as np
import torch

from . import _lib
from ._device import device, to_dev, ptr, stream_ptr
from .nmpc import _weight_matrix, _wrap_list, STATUS_TEXT


class _ArrivalCost:
    def __init__(self, model):
        self._model = model
        self.Wx = None
        self.x_guess = None
        self.Wp = None
        self.p_guess = None
        self._is_set = False

    def add_states(self, weights, guess):
        """modeling.py:747-760."""
        guess = _wrap_list(guess)
        if len(guess) != self._model.n_x:
            raise ValueError(f"The guess must have the same dimension of the model states."
                             f"There are {self._model.n_x} states but guess has {len(guess)} values.")
        self.Wx = _weight_matrix(weights, self._model.n_x, 'weights')
        self.x_guess = guess
        self._is_set = True

    def add_parameters(self, weights, guess):
        """modeling.py:762-777: (p - p_arrival)^T W (p - p_arrival) over ALL model parameters."""
        guess = _wrap_list(guess)
        if len(guess) != self._model.n_p:
            raise ValueError(f"The guess must have the same dimension of the model parameters."
                             f"There are {self._model.n_p} parameters but guess has {len(guess)} values.")
        self.Wp = _weight_matrix(weights, self._model.n_p, 'weights')
        self.p_guess = guess
        self._is_set = True


class _StageCost:
    def __init__(self, model):
        self._model = model
        self.Wy = self.Ww = None
        self.ind_y = list(range(model.n_y))
        self._is_set = False

    def add_measurements(self, weights, names=None):
        """modeling.py:686-712: the measurements `names` (default: all) enter the cost with their measurement equations.  The
        device policy evaluates the whole measurement map; a subset is the same cost with zero weight on the others, the
        measured values of the subset placed at their positions (`ind_y`)."""
        all_names = list(self._model.measurement_names)
        if names is None:
            ind = list(range(self._model.n_y))
        else:
            names = [names] if isinstance(names, str) else list(names)
            ind = []
            for n in names:
                if n not in all_names:
                    raise ValueError(f"The measurement {n} does not exist. The available states are {all_names}")
                ind.append(all_names.index(n))
        W = _weight_matrix(weights, len(ind), 'weights')
        self.Wy = np.zeros((self._model.n_y, self._model.n_y))
        self.Wy[np.ix_(ind, ind)] = W
        self.ind_y = ind
        self._is_set = True

    def add_state_noise(self, weights):
        """modeling.py:735-745."""
        self.Ww = _weight_matrix(weights, self._model.n_x, 'weights')
        self._is_set = True

    def add_inputs(self, names, weights):
        """modeling.py:714-733 adds (u - u_meas)^T W (u - u_meas) to the stage term - but the estimator builds its stage function over
        `[w, x, t_ref]` only (mhe.py:480-483) and calls it with `(w_ii, x_ii, y_ii)` (:743): the model inputs stay FREE symbols of
        that function (CasADi refuses to create it) and the input measurements are never passed; the inputs are no decision
        variables of the window either (:596-690).  The reference cannot be set up with this term - nothing to be in parity with."""
        names = [names] if isinstance(names, str) else list(names)
        for n in names:
            if n not in self._model.input_names:
                raise ValueError(f"The state {n} does not exist. The available states are {list(self._model.input_names)}")
        raise NotImplementedError("MHE stage cost on inputs: the reference's stage function leaves the model inputs as free symbols "
                                  "(mhe.py:480-483) and cannot be set up with this term; inputs are data of the window "
                                  "(add_measurements(y_meas, u_meas)), not variables")


class MovingHorizonEstimator:
    def __init__(self, model, id=None, name=None, plot_backend=None, time=0., device_index=None):
        if not model._is_setup:
            model.setup()
        self._model = model                          # continuous: collocation by default, like the reference (mhe.py:512)
        self.name = name
        self._n_x, self._n_u, self._n_p, self._n_y = model.n_x, model.n_u, model.n_p, model.n_y
        self.quad_arrival_cost = _ArrivalCost(model)
        self.quad_stage_cost = _StageCost(model)
        from .nmpc import GenericConstraint
        self.stage_constraint = GenericConstraint(model, name='stage_constraint')        # mhe.py:142
        self._horizon = None
        self._x_lb = self._x_ub = self._w_lb = self._w_ub = self._p_lb = self._p_ub = None
        self._x_guess = self._w_guess = None
        self._x_scaling = self._w_scaling = self._u_scaling = None
        self._solver_options = {}
        self._handle = None
        self._dev_index = device_index
        self._nlp_setup_done = False
        self._time = float(time)
        self._sampling_interval = model.dt
        self._meas_counter = 0
        self._horizon_is_reached = False
        self._y_hist = self._u_hist = None
        self._nlp_solution = None

    type = 'MHE'

    @property
    def horizon(self):
        return self._horizon

    @horizon.setter
    def horizon(self, n):
        if not isinstance(n, (int, np.integer)) or n <= 0:
            raise ValueError("The horizon must be a positive integer")
        self._horizon = int(n)

    def set_box_constraints(self, x_ub=None, x_lb=None, w_ub=None, w_lb=None, p_ub=None, p_lb=None, z_ub=None, z_lb=None):
        def chk(v, n, what):
            if v is None:
                return None
            v = _wrap_list(v)
            if len(v) != n:
                raise TypeError(f"The model has {n} {what}. You need to pass the same number of bounds.")
            return v
        self._x_ub, self._x_lb = chk(x_ub, self._n_x, 'states'), chk(x_lb, self._n_x, 'states')
        self._w_ub, self._w_lb = chk(w_ub, self._n_x, 'states'), chk(w_lb, self._n_x, 'states')
        self._p_ub, self._p_lb = chk(p_ub, self._n_p, 'parameters'), chk(p_lb, self._n_p, 'parameters')

    def set_initial_guess(self, x_guess=None, w_guess=None, p_guess=None, z_guess=None):
        self._x_guess = None if x_guess is None else _wrap_list(x_guess)
        self._w_guess = None if w_guess is None else _wrap_list(w_guess)
        self._p_guess = None if p_guess is None else _wrap_list(p_guess)

    def set_scaling(self, x_scaling=None, w_scaling=None, p_scaling=None, u_scaling=None):
        self._p_scaling = None if p_scaling is None else _wrap_list(p_scaling)
        self._x_scaling = None if x_scaling is None else _wrap_list(x_scaling)
        self._w_scaling = None if w_scaling is None else _wrap_list(w_scaling)
        self._u_scaling = None if u_scaling is None else _wrap_list(u_scaling)

    def set_solver_opts(self, options=None):
        self._solver_options = {k.split('.')[-1]: v for k, v in (options or {}).items()}

    def setup(self, options=None, nlp_opts=None, solver='ipopt'):
        """mhe.py:418-790."""
        if self._horizon is None:
            raise ValueError("You must set a horizon length before")
        if not (self.quad_arrival_cost._is_set or self.quad_stage_cost._is_set):
            raise ValueError("You need to define a cost function before setting up the MHE.")
        m = self._model
        opts = {'integration_method': 'discrete' if m.discrete else 'collocation', 'collocation_points': 'radau', 'degree': 3,
                'arrival_guess_update': 'smoothing', 'warm_start': True}       # optimizer.py:1410-1418 defaults
        if options is None:
            options = getattr(self, '_pending_options', None)          # set_nlp_options(...) before setup()
        for k, v in (options or {}).items():
            if k == 'integration_method' and m.discrete and v != 'discrete':
                warnings.warn(f"The integration method is set to {v} but I notice that the model is in discrete time. "
                              f"I am overwriting and using discrete mode.")
                continue
            if k == 'arrival_guess_update' and v == 'filtering':
                raise NotImplementedError("The filtering update is not yet implemented.")      # mhe.py:257-258
            opts[k] = v
        method = opts['integration_method']
        if not m.discrete:
            if method in ('rk4', 'erk'):
                # explicit Runge-Kutta inside the window: the pre-discretised model + 'discrete' 
                m = self._model = m.discretize(method) if method == 'rk4' else m.discretize('erk', order=opts.get('order', 4))
                if not m._is_setup:
                    m.setup()
                method = opts['integration_method'] = 'discrete'
            elif method == 'discrete':
                raise ValueError("integration_method 'discrete' needs a discrete-time model (Model.discretize)")
            elif method == 'multiple_shooting':
                # mhe.py:586-593, :713-718: CVODES over the interval, the state noise added to its end state.  WITHOUT state noise the
                # reference never defines the end state of the interval (`x_ii_1` is assigned inside `if
                # self._state_noise_flag`) and setup() dies with an UnboundLocalError - nothing to be in parity with.  With it, the
                # stand-in is the one the controller has for 'cvodes' (nmpc.py): a FIXED-step classic Runge-Kutta map with
                # SUNDIALS_SUBSTEPS sub-steps per sampling interval and exact derivatives of that map; no error control - a stiff
                # model needs 'collocation' (DESIGN.md 7).
                if self.quad_stage_cost.Ww is None:
                    raise NotImplementedError("integration_method 'multiple_shooting' without state noise cannot be set up in the "
                                              "reference either (mhe.py:713-718 leaves the interval's end state undefined)")
                if getattr(m, 'n_z', 0):
                    raise NotImplementedError("'multiple_shooting' on a model with algebraic states is not offloaded: use 'collocation'")
                from .nmpc import NMPC
                warnings.warn(f"integration_method 'multiple_shooting': SUNDIALS' adaptive integrator is replaced by a fixed-step "
                              f"Runge-Kutta map of order 4 with {NMPC.SUNDIALS_SUBSTEPS} sub-steps per sampling interval")
                m = self._model = m.discretize('rk4', n_sub=NMPC.SUNDIALS_SUBSTEPS)
                if not m._is_setup:
                    m.setup()
                method = opts['integration_method'] = 'discrete'
            elif method != 'collocation':
                raise NotImplementedError(f"integration method '{method}' is not offloaded")
        self._nlp_options = opts
        if nlp_opts is not None:
            self.set_solver_opts(nlp_opts)
        pinned = bool(self._n_p) and self._p_lb is not None and self._p_ub is not None and \
            list(self._p_lb) == list(self._p_ub)
        self._estimating = bool(self._n_p) and not pinned
        noise = self.quad_stage_cost.Ww is not None
        self._noise = noise
        coll = None
        if method == 'collocation':
            from .nmpc import _collocation_basis
            deg = int(opts.get('degree', 3))
            if not 1 <= deg <= 4:
                raise NotImplementedError("collocation degrees 1..4 are built")
            coll = _collocation_basis(deg, opts.get('collocation_points', 'radau'))
        self._coll = coll
        # models written as expressions, and every model under collocation: the policy is compiled around the model at setup
        # ... and every estimator WITHOUT state noise (mhe.py:599, :726-736: the collocation and discrete branches run without the
        # noise block; what tests/test_MHE.py:20-110 configure): the general policy csrc/hilo_mhe_policy.h::MheGen
        jit = bool(getattr(m, '_symbolic', False)) or coll is not None or not noise
        sc = self.stage_constraint
        if sc.is_set:
            # mhe.py:498-508, :536-553, :749-757: rows at every node and collocation point.  QUIRKS: the estimator never calls the
            # constraint's `_check_and_setup` - the expression sees the NLP's SCALED variables, and the soft branch's penalty function
            # does not exist (`self.stage_constraint.cost(e)` would call None): soft constraints cannot run in the reference
            if sc.is_soft:
                raise NotImplementedError("a soft stage constraint cannot run in the reference's estimator (its penalty function is only "
                                          "created by the controller's setup, modeling.py:839-878 is never called from mhe.py)")
            if not getattr(m, '_symbolic', False):
                raise NotImplementedError("the estimator's stage constraint needs a model written as expressions")
            jit = True
        if jit and not m.n_y:
            raise RuntimeError("The model has no measurement equations (set_measurement_equations)")
        keep = []

        def hp(a):
            if a is None:
                return None
            a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
            keep.append(a)
            return a.ctypes.data

        d = _lib.MheDesc()
        d.model_id, d.N = m.model_id, self._horizon
        if jit:
            self._user_source = m.user_source()
            if sc.is_set:
                from . import codegen
                nc = sc.size
                lbv = [-np.inf] * nc if sc.lb is None else list(sc.lb)
                ubv = [np.inf] * nc if sc.ub is None else list(sc.ub)
                if len(lbv) != nc or len(ubv) != nc:
                    raise ValueError("The dimensions of the stage constraint function and its bounds are not compatible.")
                self._user_source += codegen.mhe_fun_source(sc.constraint)
                d.n_con, d.con_lb, d.con_ub = nc, hp(lbv), hp(ubv)
            d.user_source = self._user_source.encode()
            d.user_nx, d.user_nu, d.user_np, d.user_ny = m.n_x, m.n_u, m.n_p, m.n_y
            d.user_discrete = 1 if getattr(m, '_native_discrete', False) else 0
        if coll is not None:
            d.collocation_degree = coll['d']
            d.coll_A, d.coll_D = hp(coll['A']), hp(coll['D'])
        d.erk_order = m.erk_order if m.erk_order else 4
        d.n_sub = m.n_sub
        so = self._solver_options
        d.max_iter, d.acceptable_iter = int(so.get('max_iter', 0)), int(so.get('acceptable_iter', 0))
        d.dt = m.dt
        d.tol, d.acceptable_tol = float(so.get('tol', 0.)), float(so.get('acceptable_tol', 0.))
        d.mu_init, d.bound_relax_factor = float(so.get('mu_init', 0.)), float(so.get('bound_relax_factor', -1.))
        d.Wx, d.Wy, d.Ww = hp(self.quad_arrival_cost.Wx), hp(self.quad_stage_cost.Wy), hp(self.quad_stage_cost.Ww)
        d.x_lb, d.x_ub, d.w_lb, d.w_ub = hp(self._x_lb), hp(self._x_ub), hp(self._w_lb), hp(self._w_ub)
        d.x_scaling, d.w_scaling, d.u_scaling = hp(self._x_scaling), hp(self._w_scaling), hp(self._u_scaling)
        d.x_guess, d.w_guess = hp(self._x_guess), hp(self._w_guess)
        if self._estimating:
            d.estimate_parameters = 1
            Wp = self.quad_arrival_cost.Wp
            d.Wp = hp(Wp if Wp is not None else np.zeros((self._n_p, self._n_p)))
            d.p_lb, d.p_ub = hp(self._p_lb), hp(self._p_ub)
            d.p_scaling, d.p_guess = hp(getattr(self, '_p_scaling', None)), hp(getattr(self, '_p_guess', None))
        import os
        h = C.c_void_p()
        if os.environ.get('HILO_JIT_COMPILE_ONLY'):
            # filling the run-time compiler's cache on a machine without a GPU (tools/warm_jit_cache.py): compile and stop
            if jit:
                rc = _lib.lib().hilo_mhe_create(C.byref(d), 0, C.byref(h))
                if rc != _lib.COMPILED_ONLY:
                    _lib.check(rc)
            return
        self._dev = device(self._dev_index)
        _lib.check(_lib.lib().hilo_mhe_create(C.byref(d), self._dev.index, C.byref(h)))
        self._destroy()
        self._handle = h
        N, nx, np_ = self._horizon, self._n_x, self._n_p
        dn = coll['d'] * nx if coll is not None else 0
        nw = N * nx if noise else 0                                # mhe.py:599: the noise block exists only with state noise
        ncon = sc.size if sc.is_set else 0
        self._n_v = np_ + (N + 1) * nx + nw + N * dn
        self._n_g = N * (nx + dn + ((coll['d'] if coll is not None else 0) + 1) * ncon)         # mhe.py:536-553, :728, :740, :749-757
        # bit-exact index maps of mhe.py:614-655
        self._p_ind = [list(range(np_))] if np_ else []
        self._x_ind = [list(range(np_ + k * nx, np_ + (k + 1) * nx)) for k in range(N + 1)]
        self._w_ind = [list(range(np_ + (N + 1) * nx + k * nx, np_ + (N + 1) * nx + (k + 1) * nx)) for k in range(N)] if noise else []
        off = np_ + (N + 1) * nx + nw                             # collocation states behind the noise block (mhe.py:657-671)
        self._ip_ind = [list(range(off + k * dn, off + (k + 1) * dn)) for k in range(N)] if dn else []
        self._sx = np.ones(nx) if self._x_scaling is None else np.asarray(self._x_scaling)
        self._sp = np.ones(np_) if getattr(self, '_p_scaling', None) is None else np.asarray(self._p_scaling)
        self._p_pinned = None if (not np_ or self._estimating) else to_dev(np.asarray(self._p_lb, dtype=float), self._dev, (1, -1))
        self._nlp_setup_done = True

    def _destroy(self):
        if self._handle is not None:
            _lib.lib().hilo_mhe_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    # ---- measurement ring buffer (mhe.py:274-309), kept on the device: [B, N, ny] / [B, N, nu] ---------------
    def add_measurements(self, y_meas, u_meas=None):
        if not self._nlp_setup_done:
            raise RuntimeError("You need to setup the MHE by running .setup() before running add_measurements")
        ind = self.quad_stage_cost.ind_y
        y = to_dev(y_meas, self._dev)
        ns, ny = len(ind), self._n_y
        if ns != ny:
            # the cost uses a subset of the measurements: their measured values alone (mhe.py `_check_measurements`) are placed at
            # their positions of the measurement vector; a full measurement vector is taken as it is
            width = y.shape[-1] if y.ndim >= 2 else (ns if y.numel() == ns else (ny if y.numel() % ny == 0 else ns))
            if width == ns:
                if y.numel() % ns:
                    raise ValueError(f"Dimension mismatch: {y.numel()} measured values for {ns} measurement(s) in the cost")
                ys = y.reshape(-1, ns)
                y = torch.zeros(ys.shape[0], ny, dtype=torch.float64, device=self._dev)
                y[:, ind] = ys
        y = y.reshape(-1, self._n_y)
        B = y.shape[0]
        u = None
        if self._n_u:
            if u_meas is None:
                raise ValueError(f"The model has {self._n_u} input(s); pass their measured values as u_meas")
            u = to_dev(u_meas, self._dev).reshape(-1, self._n_u)
            u = u.expand(B, -1) if u.shape[0] == 1 else u
        N = self._horizon
        if self._y_hist is None or self._y_hist.shape[0] != B:
            self._y_hist = torch.zeros(B, N, self._n_y, dtype=torch.float64, device=self._dev)
            self._u_hist = torch.zeros(B, N, max(self._n_u, 1), dtype=torch.float64, device=self._dev)
            self._meas_counter = 0
            self._horizon_is_reached = False
        if self._meas_counter < N:
            self._y_hist[:, self._meas_counter] = y
            if u is not None:
                self._u_hist[:, self._meas_counter, :self._n_u] = u
        else:                                                   # shift: the oldest sample is forgotten (mhe.py:304-309)
            self._y_hist = torch.cat([self._y_hist[:, 1:], y[:, None]], dim=1)
            if u is not None:
                self._u_hist = torch.cat([self._u_hist[:, 1:], u[:, None]], dim=1)
        self._meas_counter += 1
        if self._meas_counter >= N:
            self._horizon_is_reached = True

    def estimate(self, x_arrival=None, p_arrival=None, v0=None, runs=0, **kwargs):
        """mhe.py:311-416."""
        if not self._nlp_setup_done:
            raise ValueError("You need to setup the nlp before optimizing. Type *mheObject*.setup()")
        self._time += self._sampling_interval                   # mhe.py:333
        if not self._horizon_is_reached:
            return None, None                                   # mhe.py:415-416
        wo = getattr(self, '_warm_override', None)              # best run of a multi-start call: the next call's start vector
        self._warm_override = None
        if v0 is None and wo is not None and self._nlp_options.get('warm_start', True):
            v0 = wo
        if runs != 0:
            return self._multi_start(x_arrival, p_arrival, v0, int(runs), kwargs)
        return self._estimate_once(x_arrival, p_arrival, v0)

    def _guess_vector(self, B):
        """The initial guess in the reference's v layout [p | x_0..x_N | w_0..w_{N-1} | collocation states] (mhe.py:614-660), scaled."""
        v = np.zeros(self._n_v)
        sx = np.asarray(self._sx, dtype=float)
        xg = (np.zeros(self._n_x) if self._x_guess is None else np.asarray(self._x_guess, dtype=float)) / sx
        for ind in self._x_ind:
            v[ind] = xg
        if self._w_ind:
            sw = np.ones(self._n_x) if getattr(self, '_w_scaling', None) is None else np.asarray(self._w_scaling, dtype=float)
            wg = (np.zeros(self._n_x) if self._w_guess is None else np.asarray(self._w_guess, dtype=float)) / sw
            for ind in self._w_ind:
                v[ind] = wg
        for ind in getattr(self, '_ip_ind', None) or []:
            v[ind] = np.tile(xg, len(ind) // self._n_x)
        if self._n_p and self._p_ind:
            pg = getattr(self, '_p_guess', None)
            sp = np.ones(self._n_p) if getattr(self, '_sp', None) is None else np.asarray(self._sp, dtype=float)
            v[self._p_ind[0]] = (np.zeros(self._n_p) if pg is None else np.asarray(pg, dtype=float)) / sp
        return to_dev(np.tile(v, (B, 1)), self._dev)

    def _multi_start(self, x_arrival, p_arrival, v0, runs, kwargs):
        """mhe.py:386-399: `runs` solves of the SAME window, the first from the given start (the caller's v0, else the previous
        solution, else the initial guess), the following from `v0 + v0 (1 - 2 rand) pert_factor`; per instance the solve with the
        smallest objective is kept (the reference compares `f` alone, mhe.py:392) and becomes the next call's warm start (:398).
        The reference draws from the unseeded numpy generator; here `seed=` (default 0) makes the draws reproducible, like
        `NMPC.optimize(runs=)`."""
        pert = float(kwargs.get('pert_factor', 0.1))
        gen = torch.Generator(device='cpu').manual_seed(int(kwargs.get('seed', 0)))
        B = self._y_hist.shape[0]
        prev = self._nlp_solution
        if v0 is not None:
            base = to_dev(v0, self._dev).reshape(-1, self._n_v)
            base = (base.expand(B, -1) if base.shape[0] == 1 else base).clone()
        elif prev is not None and prev['x'].shape[0] == B and self._nlp_options.get('warm_start', True):
            base = prev['x'].clone()
        else:
            base = self._guess_vector(B)
        # (every run sees the same arrival values: those of the call, else of the state BEFORE the first run)
        xa_fix = x_arrival if x_arrival is not None else (prev['x'][:, self._x_ind[2]].clone() if prev is not None and prev['x'].shape[0] == B else None)
        pa_fix = p_arrival if (p_arrival is not None or not self._estimating) else \
            (prev['x'][:, self._p_ind[0]].clone() if prev is not None and prev['x'].shape[0] == B else None)
        best, bx, bp = None, None, None
        start = v0 if v0 is not None else base
        for _ in range(runs):
            self._nlp_solution = prev if xa_fix is None else self._nlp_solution
            x_opt, p_opt = self._estimate_once(xa_fix, pa_fix, start)
            sol = self._nlp_solution
            if best is None:
                best = {k: v.clone() for k, v in sol.items()}
                bx, bp = x_opt.clone(), (None if p_opt is None else p_opt.clone())
            else:
                take = sol['f'] < best['f']
                for k in sol:
                    best[k][take] = sol[k][take]
                bx[take] = x_opt[take]
                if bp is not None:
                    bp[take] = p_opt[take]
            rnd = torch.rand(base.shape, generator=gen, dtype=torch.float64).to(self._dev)
            start = base + base * (1 - 2 * rnd) * pert
        self._nlp_solution = best
        self._warm_override = best['x']
        return bx, bp

    def _estimate_once(self, x_arrival, p_arrival, v0):
        B = self._y_hist.shape[0]
        dev = self._dev
        if x_arrival is not None:
            xa = to_dev(x_arrival, dev).reshape(-1, self._n_x)
            if xa.shape[1] != self._n_x:
                raise ValueError(f'The model has {self._n_x} states(s): {self._model.dynamical_state_names}. You must '
                                 f'pass me a guess of the values before running the optimization')
            xa = xa.expand(B, -1) if xa.shape[0] == 1 else xa
        elif self._nlp_solution is not None and self._nlp_solution['x'].shape[0] == B:
            # 'smoothing' update: x_2 of the previous solution (mhe.py:254-256) - taken as stored, i.e. scaled
            xa = self._nlp_solution['x'][:, self._x_ind[2]]
        else:
            g = self.quad_arrival_cost.x_guess
            xa = to_dev(np.tile(np.asarray(g if g is not None else np.zeros(self._n_x), dtype=float), (B, 1)), dev)
        xa = xa.contiguous()
        v0t = None
        if v0 is not None:
            v0t = to_dev(v0, dev).reshape(-1, self._n_v)
            v0t = (v0t.expand(B, -1) if v0t.shape[0] == 1 else v0t).contiguous()
        elif not self._nlp_options.get('warm_start', True):
            _lib.check(_lib.lib().hilo_mhe_reset_warm_start(self._handle))
        p, ps = (self._p_pinned, 0) if self._n_p else (None, 0)
        if self._estimating:
            # `p` = arrival value of the estimated parameters and value of the pinned ones (mhe.py:352-356): given, else
            # the previous estimate (scaled, like the state smoothing of mhe.py:254-256), else the arrival-cost guess
            if p_arrival is not None:
                pa = to_dev(p_arrival, dev).reshape(-1, self._n_p)
            elif self._nlp_solution is not None and self._nlp_solution['x'].shape[0] == B:
                pa = self._nlp_solution['x'][:, self._p_ind[0]]
            else:
                g = self.quad_arrival_cost.p_guess
                if g is None:
                    g = getattr(self, '_p_guess', None) or np.zeros(self._n_p)
                pa = to_dev(np.asarray(g, dtype=float), dev).reshape(1, -1)
            pa = (pa.expand(B, -1) if pa.shape[0] == 1 else pa).contiguous()
            p, ps = pa, self._n_p
        v_opt = torch.empty(B, self._n_v, dtype=torch.float64, device=dev)
        f_opt = torch.empty(B, dtype=torch.float64, device=dev)
        lam_g = torch.empty(B, self._n_g, dtype=torch.float64, device=dev)
        x_opt = torch.empty(B, self._n_x, dtype=torch.float64, device=dev)
        status = torch.empty(B, dtype=torch.int32, device=dev)
        iters = torch.empty(B, dtype=torch.int32, device=dev)
        kkt = torch.empty(B, dtype=torch.float64, device=dev)
        u_hist = self._u_hist[:, :, :self._n_u].contiguous() if self._n_u else None
        _lib.check(_lib.lib().hilo_mhe_estimate(self._handle, B, ptr(xa), ptr(p), ps, ptr(u_hist),
                                                ptr(self._y_hist.contiguous()), ptr(v0t), ptr(v_opt), ptr(f_opt),
                                                ptr(lam_g), ptr(x_opt), ptr(status), ptr(iters), ptr(kkt),
                                                stream_ptr(dev)))
        self._nlp_solution = {'x': v_opt, 'f': f_opt, 'lam_g': lam_g, 'status': status, 'iter_count': iters,
                              'kkt_error': kkt}
        if not self._n_p:
            p_opt = None
        elif self._estimating:
            p_opt = v_opt[:, self._p_ind[0]] * torch.as_tensor(self._sp, device=dev)      # mhe.py:377-378
        else:
            p_opt = self._p_pinned.expand(B, -1)
        return x_opt, p_opt

    def return_mhe_estimation(self):
        """mhe.py:1214-1232: (x_pred [B, n_x, N+1], w_pred [B, n_x, N]) of the last estimate, un-scaled."""
        if self._nlp_solution is None:
            warnings.warn("There is still no mpc solution available. Run mpc.optimize() to get one.")
            return None, None
        v = self._nlp_solution['x'].cpu().numpy()
        N, nx = self._horizon, self._n_x
        sw = np.ones(nx) if getattr(self, '_w_scaling', None) is None else np.asarray(self._w_scaling, dtype=float)
        X = v[:, self._x_ind[0][0]:self._x_ind[N][-1] + 1].reshape(-1, N + 1, nx) * self._sx
        if not self._w_ind:
            return np.swapaxes(X, 1, 2), None
        W = v[:, self._w_ind[0][0]:self._w_ind[N - 1][-1] + 1].reshape(-1, N, nx) * sw
        return np.swapaxes(X, 1, 2), np.swapaxes(W, 1, 2)

    @property
    def has_state_noise(self):
        """mhe.py:1234-1246 (this build offloads the state-noise variant only)."""
        return self.quad_stage_cost.Ww is not None

    @has_state_noise.setter
    def has_state_noise(self, arg):
        if not isinstance(arg, bool):
            raise TypeError("has_state_noise accepts True or False")
        if arg and self.quad_stage_cost.Ww is None:
            raise ValueError("state noise needs its weights: quad_stage_cost.add_state_noise(weights=...)")
        if not arg:
            self.quad_stage_cost.Ww = None

    def set_nlp_options(self, *args, **kwargs):
        """mhe.py:792-860: the options `setup(options=...)` takes, checked against the reference's allow-lists."""
        possible = {'integration_method': ['collocation', 'rk4', 'erk', 'discrete', 'multiple_shooting'],
                    'collocation_points': ['radau', 'legendre'], 'degree': None, 'print_level': [0, 1],
                    'arrival_guess_update': ['filtering', 'smoothing'], 'warm_start': [True, False]}
        given = args[0] if (args and isinstance(args[0], dict)) else kwargs
        for k, v in (given or {}).items():
            if k not in possible:
                raise ValueError(f"The option named {k} does not exist. Possible options are {list(possible)}.")
            if possible[k] is not None and v not in possible[k]:
                raise ValueError(f"The option {k} is set to value {v} but the only allowed values are {possible[k]}.")
        self._pending_options = dict(given or {})

    def set_time_varying_parameters(self, time_varying_parameters=None):
        """mhe.py:911-932."""
        if time_varying_parameters:
            for tvp in time_varying_parameters:
                if tvp not in self._model.parameter_names:
                    raise ValueError(f"The time-varying parameter {tvp} is not in the model0 parameter. "
                                     f"The model0 parameters are {self._model.parameter_names}.")
            # the reference declares a `tv_p` slot in the solver's parameter struct (mhe.py:677) and never reads it: `p` stays the one
            # decision vector of the window (mhe.py:614-623, :726-736), `estimate()` fills no values in.  Same here: the names are
            # checked and remembered, the NLP is unchanged.
            self._time_varying_parameters = list(time_varying_parameters)
            warnings.warn("time-varying parameters are declared, but - like in the reference, whose estimator never reads its tv_p "
                          "slot (mhe.py:677) - they do not enter the estimation problem: p is one vector for the whole window")

    def set_aux_nonlinear_constraints(self, aux_nl_const=None, ub=None, lb=None):
        """mhe.py:1070-1087."""
        if None not in [aux_nl_const, ub, lb]:
            raise NotImplementedError("nonlinear constraints inside the estimation window are not offloaded")
        if not all(a is None for a in (aux_nl_const, ub, lb)):
            raise ValueError("When passing nonlinear constraints, you must pass"
                             "the nonlinear constraint function, lower and upper bound")

    @property
    def solver_status_code(self):
        return None if self._nlp_solution is None else self._nlp_solution['status'].cpu().numpy()

    def stats(self):
        s = self._nlp_solution
        if s is None:
            return {}
        st = s['status'].cpu().numpy()
        return {'return_status': [STATUS_TEXT.get(int(c), 'other') for c in st], 'success': (st == 1) | (st == 2),
                'iter_count': s['iter_count'].cpu().numpy(), 'kkt_error': s['kkt_error'].cpu().numpy()}


MHE = MovingHorizonEstimator
