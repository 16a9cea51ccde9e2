"""Batched linear MPC on the GPU.

API mirror of `hilo_mpc.LMPC` (hilo_mpc/modules/controller/mpc.py:1975-2460): `Q`, `R`, `P`, `horizon`,
`set_box_constraints`, `set_scaling`, `setup(solver=...)`, `optimize(x0)` with a leading batch axis on `x0`.
The QP is assembled exactly as `LMPC.setup` does (mpc.py:2198-2266) - including the `kron(B, I_N)` input block of the
parameter-free branch (mpc.py:2243), selectable with `setup(kron_variant='reference'|'corrected')` - and
solved by `hilo_qp_solve` (dense interior point) in libhilo_hip.so.  Cost convention 1/2 v^T H v (mpc.py:2374, Q6).
"""
import ctypes as C
import warnings

import numpy as np
import torch

from . import _lib
from ._device import device, to_dev, ptr, stream_ptr
from .nmpc import _wrap_list, STATUS_TEXT


class LMPC:
    _solver_name_list_qp = ['qpoases', 'hip_qp']

    def __init__(self, model, id=None, name=None, plot_backend=None, device_index=None):
        sym = getattr(model, '_symbolic', False)
        if model.name != 'lti' and not model.is_linear():
            raise TypeError("The model is nonlinear. Use the NMPC class or linearize the model.")
        if model.name != 'lti' and not sym:
            raise NotImplementedError("LMPC needs the matrices of the system: Model('lti', A=..., B=...) or a linear / linearised "
                                      "model written as expressions")
        if sym and not model.discrete:
            raise NotImplementedError("LMPC predicts with x+ = A x + B u: discretize the model first (Model.discretize)")
        if not model._is_setup:
            model.setup()
        self._model = model
        self._n_x, self._n_u = model.n_x, model.n_u
        self._Q = self._R = self._P = None
        self._horizon = None
        self._x_lb = self._x_ub = self._u_lb = self._u_ub = None
        self._x_scaling = self._u_scaling = None
        self._handle = None
        self._dev_index = device_index
        self._time = 0.
        self._n_iterations = 0
        self._nlp_solution = None
        self._sampling_interval = model.dt
        self._time_varying_parameters, self._time_varying_parameters_values, self._tvp_window = [], None, None

    type = 'LMPC'

    def _mat(self, arg, n):
        a = np.atleast_2d(np.asarray(arg, dtype=float))
        if a.shape != (n, n):
            raise ValueError(f"weight matrix must be {n}x{n}, got {a.shape}")
        return a

    Q = property(lambda s: s._Q)
    R = property(lambda s: s._R)
    P = property(lambda s: s._P)

    @Q.setter
    def Q(self, arg):
        self._Q = self._mat(arg, self._n_x)

    @R.setter
    def R(self, arg):
        self._R = self._mat(arg, self._n_u)

    @P.setter
    def P(self, arg):
        self._P = self._mat(arg, self._n_x)

    @property
    def horizon(self):
        return self._horizon

    @horizon.setter
    def horizon(self, n):
        if not isinstance(n, (int, np.integer)) or n <= 0:
            raise ValueError("The horizon must be a positive integer")
        self._horizon = int(n)

    prediction_horizon = horizon
    control_horizon = horizon

    # mpc.py:2396-2406: not available for the linear MPC
    def set_stage_constraints(self, *args, **kwargs):
        raise NotImplementedError("The method set_stage_constraints is not available for LMPC.")

    def set_custom_constraints_function(self, *args, **kwargs):
        raise NotImplementedError("The method set_custom_constraints_function is not available for LMPC.")

    def set_initial_guess(self, *args, **kwargs):
        raise NotImplementedError("The method set_initial_guess is not available for LMPC.")

    def set_box_constraints(self, x_ub=None, x_lb=None, u_ub=None, u_lb=None):
        def chk(v, n, what):
            if v is None:
                return None
            v = _wrap_list(v)
            if len(v) != n:
                raise TypeError(f"The model has {n} {what}. You need to pass the same number of bounds.")
            return v
        self._x_ub, self._x_lb = chk(x_ub, self._n_x, 'states'), chk(x_lb, self._n_x, 'states')
        self._u_ub, self._u_lb = chk(u_ub, self._n_u, 'inputs'), chk(u_lb, self._n_u, 'inputs')

    def set_scaling(self, x_scaling=None, u_scaling=None):
        self._x_scaling = None if x_scaling is None else _wrap_list(x_scaling)
        self._u_scaling = None if u_scaling is None else _wrap_list(u_scaling)

    def set_time_varying_parameters(self, names=None, values=None):
        """optimizer.py:1519-1560 (see NMPC.set_time_varying_parameters): parameters of the model whose value changes along the
        horizon - the prediction then uses A(p_k), B(p_k) per stage (mpc.py:2200-2206, :2236-2240)."""
        if names is None:
            names = []
        if not (isinstance(names, (list, tuple)) and all(isinstance(n, str) for n in names)):
            raise ValueError('Tvp must be a list of strings with the paramers name that are time varying')
        for tvp in names:
            if tvp not in self._model.parameter_names:
                raise ValueError(f"I could not find the parameter {tvp} in the model. The models parameters are "
                                 f"{self._model.parameter_names}.")
        if values is not None:
            if not isinstance(values, dict):
                raise TypeError("The values parameter must be a dictionary.")
            for key in values:
                if key not in names:
                    raise ValueError(f"The key {key} is not in the name vector: {names}. You need to pass a dictionary "
                                     f"where the keys are the name of the time varying parameters.")
        self._time_varying_parameters, self._time_varying_parameters_values, self._tvp_window = list(names), values, None

    n_tvp = property(lambda s: len(s._time_varying_parameters))

    def _n_par(self):
        return self._model.n_p if getattr(self._model, '_symbolic', False) else 0

    def _equality_matrix(self, As, Bs, kron_variant):
        """Aeq of mpc.py:2198-2245 from the stage matrices: one (A, B) for all stages - `kron(I, A)` and the input block of the
        parameter-free branch, `kron(B, I_N)` (:2243) or its corrected form - or one pair per stage (`diagcat`,
        :2200-2206, :2236-2240)."""
        N, nx, nu = self._horizon, self._n_x, self._n_u
        aux2 = np.zeros((N, N + 1))
        for i in range(N):
            aux2[i, i + 1] = -1
        Abar2 = np.kron(aux2, np.eye(nx))                                                         # mpc.py:2233
        if len(As) == 1:
            Abar1 = np.kron(np.eye(N), As[0])                                                     # mpc.py:2209-2210
            Abar3 = np.kron(Bs[0], np.eye(N)) if kron_variant == 'reference' else np.kron(np.eye(N), Bs[0])   # mpc.py:2243
        else:
            Abar1, Abar3 = np.zeros((N * nx, N * nx)), np.zeros((N * nx, N * nu))
            for k in range(N):
                Abar1[k * nx:(k + 1) * nx, k * nx:(k + 1) * nx] = As[k]
                Abar3[k * nx:(k + 1) * nx, k * nu:(k + 1) * nu] = Bs[k]
        Abar1 = np.hstack([Abar1, np.zeros((N * nx, nx))])                                        # mpc.py:2213
        return np.hstack([Abar1 + Abar2, Abar3])                                                  # mpc.py:2245

    def _stage_parameters(self, cp, tvp):
        """[N][n_p] parameter values along the horizon: constant ones from `cp`, time-varying ones from `tvp` or the stored series
        (window advancing with the iteration counter, mpc.py:292-333, :2013-2045)."""
        N, names, tv = self._horizon, self._model.parameter_names, self._time_varying_parameters
        npar, n_tvp = len(names), len(tv)
        cpv = np.zeros(0) if cp is None else np.asarray(cp.cpu() if isinstance(cp, torch.Tensor) else cp, dtype=float).ravel()
        if cpv.size != npar - n_tvp:
            raise ValueError(f"The model has {npar - n_tvp} constant parameter(s): {[n for n in names if n not in tv]}. You must "
                             f"pass me the value of these before running the optimization to the 'cp' parameter.")
        win = np.zeros((n_tvp, N))
        if n_tvp:
            ci = self._n_iterations
            if tvp is not None:
                for key, value in tvp.items():
                    if key not in tv:
                        raise ValueError(f"The parameter {key} was not declared time varying (set_time_varying_parameters: {tv})")
                    r = tv.index(key)                       # rows of the window follow the declared names, not the dictionary's order
                    if len(value) < N:
                        raise TypeError(f"When passing time-varying parameters, you need to pass a number of values at least as "
                                        f"long as the prediction horizon. The parameter {key} has {len(value)} values but the MPC "
                                        f"has a prediction horizon length of {N}.")
                    win[r] = np.asarray(value[0:N], dtype=float)
            elif self._time_varying_parameters_values is not None:
                vals = self._time_varying_parameters_values
                if ci == 0 or self._tvp_window is None:
                    for r, name in enumerate(tv):
                        if len(vals[name]) < N:
                            raise TypeError(f"The parameter {name} has {len(vals[name])} values but the MPC has a prediction "
                                            f"horizon length of {N}.")
                        win[r] = np.asarray(vals[name][0:N], dtype=float)
                else:
                    win = self._tvp_window.copy()
                    win[:, :-1] = win[:, 1:]
                    for r, name in enumerate(tv):
                        value = vals[name]
                        if ci + N > len(value):
                            warnings.warn("The prediction horizon is predicting outside the values of the time varying "
                                          "parameters. I am now taking looping back the values and start from there.")
                            win[r, -1] = value[ci - N * int(np.floor(ci / N))]
                        else:
                            win[r, -1] = value[ci + N - 1]
                self._tvp_window = win
            else:
                raise ValueError(f"Mate, I know there are {n_tvp} time varying parameters but you did not pass me any."
                                 f"Please provide me with the values of the parameters, either to the optimize() method or to "
                                 f"the set_time_varying_parameters() method.")
        P = np.empty((N if n_tvp else 1, npar))
        for k in range(P.shape[0]):
            ic = 0
            for j, name in enumerate(names):
                if name in tv:
                    P[k, j] = win[tv.index(name), k]        # by NAME: the declared order need not be the model's parameter order
                else:
                    P[k, j] = cpv[ic]
                    ic += 1
        return P

    def setup(self, options=None, solver_options=None, solver='qpoases', kron_variant='reference'):
        """mpc.py:2143-2305.  `kron_variant='reference'` reproduces mpc.py:2243 (`kron(B, I_N)`), 'corrected' uses the
        block-diagonal input matrix of the time-varying branch (mpc.py:2236-2240)."""
        if solver not in self._solver_name_list_qp:
            raise ValueError(f"The solver {solver} does no exist. The possible solver are {self._solver_name_list_qp}.")
        if self._horizon is None:
            raise ValueError("You must set a prediction horizon length before")
        if kron_variant not in ('reference', 'corrected'):
            raise ValueError("kron_variant must be 'reference' or 'corrected'")
        N, nx, nu = self._horizon, self._n_x, self._n_u
        # mpc.py:2183-2184: `state_matrix`, `input_matrix` of the model - of a model written as expressions the Jacobians of its
        # (discretised) equations at the equilibrium point.  With parameters the matrices are formed per call from `cp` / `tvp`
        # (mpc.py:2343-2366 substitutes them into Aeq there as well); setup() then only fixes the sizes.
        self._kron_variant = kron_variant
        self._aeq_key = None
        if self._n_par():
            A, B = np.zeros((nx, nx)), np.zeros((nx, nu))
        else:
            if self._time_varying_parameters:
                raise ValueError("time-varying parameters were declared, but the model has no parameters")
            A, B, _ = self._model.system_matrices()
        Q = np.zeros((nx, nx)) if self._Q is None else self._Q                                  # mpc.py:2188-2193
        P = np.zeros((nx, nx)) if self._P is None else self._P
        R = np.zeros((nu, nu)) if self._R is None else self._R
        sx = np.ones(nx) if self._x_scaling is None else np.asarray(self._x_scaling)
        su = np.ones(nu) if self._u_scaling is None else np.asarray(self._u_scaling)
        Aeq = self._equality_matrix([A], [B], kron_variant)
        n_v = (N + 1) * nx + N * nu
        H = np.zeros((n_v, n_v))                                                                  # mpc.py:2252-2256
        H[:N * nx, :N * nx] = np.kron(np.eye(N), Q)
        H[N * nx:(N + 1) * nx, N * nx:(N + 1) * nx] = P
        H[(N + 1) * nx:, (N + 1) * nx:] = np.kron(np.eye(N), R)
        inf = np.inf
        xl = (np.full(nx, -inf) if self._x_lb is None else np.asarray(self._x_lb, dtype=float)) / sx   # mpc.py:2071-2075
        xu = (np.full(nx, inf) if self._x_ub is None else np.asarray(self._x_ub, dtype=float)) / sx
        ul = (np.full(nu, -inf) if self._u_lb is None else np.asarray(self._u_lb, dtype=float)) / su
        uu = (np.full(nu, inf) if self._u_ub is None else np.asarray(self._u_ub, dtype=float)) / su
        self._dev = device(self._dev_index)
        self._H = to_dev(H, self._dev)
        self._g = torch.zeros(n_v, dtype=torch.float64, device=self._dev)                        # mpc.py:2258
        self._Ad = to_dev(Aeq, self._dev)
        self._beq = torch.zeros(N * nx, dtype=torch.float64, device=self._dev)                   # mpc.py:2248
        self._v_lb = to_dev(np.concatenate([np.tile(xl, N + 1), np.tile(ul, N)]), self._dev)      # mpc.py:2259-2266
        self._v_ub = to_dev(np.concatenate([np.tile(xu, N + 1), np.tile(uu, N)]), self._dev)
        self._x_ind = [list(range(k * nx, (k + 1) * nx)) for k in range(N + 1)]                 # mpc.py:2221-2231
        self._u_ind = [list(range((N + 1) * nx + k * nu, (N + 1) * nx + (k + 1) * nu)) for k in range(N)]
        self._sx, self._su = to_dev(sx, self._dev), to_dev(su, self._dev)
        self._unit_sx, self._unit_su = bool(np.all(sx == 1.)), bool(np.all(su == 1.))
        self._bound_rows = None
        self._n_v, self._n_g = n_v, N * nx
        h = C.c_void_p()
        _lib.check(_lib.lib().hilo_qp_create(n_v, N * nx, self._dev.index, C.byref(h)))
        so = {k.split('.')[-1]: v for k, v in (solver_options or {}).items()}
        _lib.check(_lib.lib().hilo_qp_set_options(h, float(so.get('tol', 0.)), int(so.get('max_iter', 0))))
        self._pinned_box = bool(np.any(xl == xu) or np.any(ul == uu))
        self._destroy()
        self._handle = h
        self._declare_stages(Aeq, [A], [B])

    def _declare_stages(self, Aeq, As, Bs):
        """A QP with the stage shape x_{k+1} = A_k x_k + B_k u_k - the time-varying branch (mpc.py:2236-2240), the corrected input
        block, or sizes where `kron(B, I_N)` (:2243) happens to be it - is solved stage by stage (csrc/hilo_qp_ocp.h, which reads
        ONLY the block-diagonal positions of the input block); the reference's `kron(B, I_N)` couples the stages differently and
        stays on the dense kernels, and so do variables pinned by equal bounds other than x_0.  Decided from the equality block
        itself, every time it is assembled (setup, new parameter values)."""
        staged = len(As) > 1 or np.array_equal(Aeq, self._equality_matrix(As, Bs, 'corrected'))
        staged = staged and not self._pinned_box
        used = C.c_int(0)
        _lib.check(_lib.lib().hilo_qp_set_stages(self._handle, self._n_x, self._n_u, self._horizon if staged else 0, C.byref(used)))
        self._qp_stages = bool(used.value)

    def _destroy(self):
        if self._handle is not None:
            _lib.lib().hilo_qp_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def optimize(self, x0, tvp=None, cp=None):
        """mpc.py:2307-2394."""
        if self._handle is None:
            raise ValueError("Howdy! You need to setup the MPC before optimizing. Run .setup() on the MPC object.")
        if self._n_par():
            P = self._stage_parameters(cp, tvp)
            # Aeq also depends on the equilibrium point the matrices are taken at and on the sampling interval
            # (mpc.py:2350-2353 substitutes x_eq / u_eq with every call)
            eq = [np.asarray(getattr(self._model, a, None) if getattr(self._model, a, None) is not None else [], dtype=float).ravel()
                  for a in ('_x_eq', '_u_eq')]
            key = P.tobytes() + b'|' + eq[0].tobytes() + b'|' + eq[1].tobytes() + b'|' + repr(self._model.dt).encode()
            if key != self._aeq_key:        # new parameter values: the equality block of this QP (host assembly, one upload)
                # one pair per stage only with time-varying parameters (`diagcat`, mpc.py:2200-2206, :2236-2240); constant
                # parameters alone leave the reference on its `kron(I, A)` / `kron(B, I_N)` branch (:2208-2210, :2241-2243)
                mats = [self._model.system_matrices(p=pk) for pk in (P if self._time_varying_parameters else P[:1])]
                As, Bs = [m[0] for m in mats], [m[1] for m in mats]
                Aeq = self._equality_matrix(As, Bs, self._kron_variant)
                self._Ad, self._aeq_key = to_dev(Aeq, self._dev), key
                self._declare_stages(Aeq, As, Bs)
        else:
            if tvp is not None:
                raise ValueError("time-varying parameter values were passed, but the model has no parameters")
            if cp is not None:
                warnings.warn("You are passing a parameter vector in the optimizer, but the model has no defined "
                              "parameters. I am ignoring the vector.")
        host = not isinstance(x0, torch.Tensor)
        x = to_dev(x0, self._dev)
        single = x.ndim <= 1 or (x.ndim == 2 and x.shape[1] == 1 and x.shape[0] == self._n_x and self._n_x != 1)
        x = x.reshape(1, -1) if single else x
        if x.shape[1] != self._n_x:
            raise ValueError(f"We have an issue mate, the x0 you supplied has dimension {x.shape[1]} but the model has "
                             f"{self._n_x} states.")
        B, dev, n, m = x.shape[0], self._dev, self._n_v, self._n_g
        # mpc.py:2361-2362 writes the measured state into lbx / ubx before every solver call; here the bound rows are ONE pair shared
        # by the batch and the solve takes the pinned values themselves (hilo_qp_solve_pinned): one launch per step
        xs = (x if self._unit_sx else x / self._sx).contiguous()
        # the result vectors of a call: two allocations, contiguous views (a call at this size is bound by the host's time)
        buf = torch.empty(B * (2 * n + m + 1), dtype=torch.float64, device=dev)
        v, lam_x, lam_a, f = buf[:B * n].view(B, n), buf[B * n:2 * B * n].view(B, n), buf[2 * B * n:B * (2 * n + m)].view(B, m), \
            buf[B * (2 * n + m):]
        ibuf = torch.empty(2, B, dtype=torch.int32, device=dev)
        status, iters = ibuf[0], ibuf[1]
        _lib.check(_lib.lib().hilo_qp_solve_pinned(self._handle, B, ptr(self._H), 0, ptr(self._g), 0, ptr(self._Ad), 0, ptr(self._v_lb),
                                                   ptr(self._v_ub), 0, ptr(xs), self._n_x, self._n_x, ptr(self._beq), ptr(self._beq), 0,
                                                   ptr(v), ptr(f), ptr(lam_a), ptr(lam_x), ptr(status), ptr(iters), stream_ptr(dev)))
        self._nlp_solution = {'x': v, 'f': f, 'lam_a': lam_a, 'lam_x': lam_x, 'status': status, 'iter_count': iters}
        self._time += self._sampling_interval                                                     # mpc.py:2386
        self._n_iterations += 1                                                                   # mpc.py:2392
        u = v[:, self._u_ind[0][0]:self._u_ind[0][-1] + 1]                                        # mpc.py:2377
        if not self._unit_su:
            u = u * self._su                   # (unit scaling: a view of this call's own result vector, no copy kernel)
        if host:
            u = u.cpu().numpy()
            return u.reshape(-1, 1) if single else u
        return u[0] if single else u

    @property
    def solver_status_code(self):
        return None if self._nlp_solution is None else self._nlp_solution['status'].cpu().numpy()

    def stats(self):
        s = self._nlp_solution
        if s is None:
            return {}
        st = s['status'].cpu().numpy()
        return {'return_status': [STATUS_TEXT.get(int(c), 'other') for c in st], 'success': st == 1,
                'iter_count': s['iter_count'].cpu().numpy()}

    def return_prediction(self):
        v = self._nlp_solution['x'].cpu().numpy()
        N, nx, nu = self._horizon, self._n_x, self._n_u
        X = v[:, :(N + 1) * nx].reshape(-1, N + 1, nx) * self._sx.cpu().numpy()
        U = v[:, (N + 1) * nx:].reshape(-1, N, nu) * self._su.cpu().numpy()
        return np.swapaxes(X, 1, 2), np.swapaxes(U, 1, 2)
