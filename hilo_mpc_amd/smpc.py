"""Stochastic NMPC: `SMPC` of the reference (hilo_mpc/modules/controller/mpc.py:2462-2808) behind the same interface.

The reference turns the stochastic problem  x+ = f(x, u) + B_w (g(x) + w),  g a trained Gaussian process, into a deterministic
NMPC on a SURROGATE model (`_create_deterministic_surrogate`, :2512-2614): the states are the mean of x and the entries of its
covariance matrix Kx (column-major, n_x^2 extra states),

    mean+ = f(mean, u) + B_w gp_mean(mean)
    Kx+   = [df/d(x,u)  B_w] bigK [df/d(x,u)  B_w]^T,   bigK = [[Kz, Kz jgp^T], [jgp Kz, Kd0 + jgp Kz jgp^T]],
    Kz    = [[Kx, Kx K^T], [K Kx, K Kx K^T]]            (K the gain of the ancillary controller),

with jgp the Jacobian of the GP mean and Kd0 its posterior variance; box constraints become chance constraints
`x + sqrt(2) erfinv(2 p - 1) sqrt(Kx_ii + 1e-8) <= x_ub` per stage and at the end of the horizon (:2623-2645), and the cost gets
`trace(Q Kx) + trace(R Ku)` (:2766-2771).  Everything else is `NMPC`.

Here the surrogate is built from the model's expressions (hilo_mpc_amd/expr.py: `jacobian` for df/d(x,u); the learned terms
become `gp` / `gpd` / `gpvar` nodes: posterior mean, its derivative with respect to a feature and the posterior variance,
csrc/hilo_models.h::gp_se_mean / gp_se_dmean / gp_se_var) and compiled at `setup()` like any other model written as
expressions; the solve is the general run-time compiled policy of `NMPC` (csrc/hilo_nmpc_user.h).
"""
import math

import numpy as np
import torch

from ._device import to_dev
from .expr import Expr, _add, _mul, jacobian, sqrt
from .model import Model
from .nmpc import NMPC, _wrap_list


def _erfinv(y):
    """scipy.special.erfinv (mpc.py:31) without the dependency: Newton's method to round-off - on erf in the centre, on erfc
    (1 - |y| is exact in floating point there) in the tails, where erf(x) - y would cancel."""
    y = float(y)
    if y <= -1.0:
        return -math.inf
    if y >= 1.0:
        return math.inf
    a = abs(y)
    t = 1.0 - a
    w = -math.log(t * (1.0 + a))
    x = math.sqrt(w) * 0.8 if w > 6.25 else a * 0.8862269254527580     # rough start, Newton converges from either
    c = 2.0 / math.sqrt(math.pi)
    for _ in range(100):
        err = (math.erf(x) - a) if a <= 0.5 else (t - math.erfc(x))
        step = err / (c * math.exp(-x * x))
        x -= step
        if abs(step) <= 1e-16 * max(1.0, abs(x)):
            break
    return math.copysign(x, y)


# explicit Runge-Kutta tableaux reachable through `discretize('erk', order=...)` / 'rk4' (modeling.py:1008-1085, :1239-1250)
_ERK = {1: ([[0.]], [1.]),
        2: ([[0., 0.], [.5, 0.]], [0., 1.]),
        3: ([[0., 0., 0.], [.5, 0., 0.], [-1., 2., 0.]], [1 / 6, 2 / 3, 1 / 6]),
        4: ([[0., 0., 0., 0.], [.5, 0., 0., 0.], [0., .5, 0., 0.], [0., 0., 1., 0.]], [1 / 6, 1 / 3, 1 / 3, 1 / 6])}


def _c(v):
    return Expr.wrap(float(v))


def _matmul(A, B):
    """Product of matrices of expressions (lists of rows) with zero / unit factors removed."""
    n, k, m = len(A), len(B), len(B[0]) if B else 0
    out = [[_c(0.0) for _ in range(m)] for _ in range(n)]
    for i in range(n):
        for j in range(m):
            acc = _c(0.0)
            for q in range(k):
                acc = _add(acc, _mul(A[i][q], B[q][j]))
            out[i][j] = acc
    return out


def _T(A):
    return [list(r) for r in zip(*A)] if A else []


def _hcat(*Ms):
    return [sum((list(M[i]) for M in Ms), []) for i in range(len(Ms[0]))]


def _vcat(*Ms):
    return [list(r) for M in Ms for r in M]


def _madd(A, B):
    return [[_add(a, b) for a, b in zip(ra, rb)] for ra, rb in zip(A, B)]


class SMPC(NMPC):
    """Class for Stochastic Nonlinear Model Predictive Control (mpc.py:2462)."""
    type = 'SMPC'

    def __init__(self, det_model, stoch_model, B, id=None, name=None, plot_backend=None, use_sx=True, stats=False, Kgain=None,
                 device_index=None):
        self._n_x_s, self._n_u_s, self._n_p_s = det_model.n_x, det_model.n_u, det_model.n_p
        self._n_y_s, self._n_z_s = det_model.n_y, getattr(det_model, 'n_z', 0)
        self._Kgain_is_set = Kgain is not None
        if Kgain is not None:
            Kgain = np.atleast_2d(np.asarray(Kgain, dtype=float))
            if Kgain.shape != (det_model.n_u, det_model.n_x):
                raise ValueError(f"Kgain must have shape ({det_model.n_u}, {det_model.n_x})")
        model_c, Kx, Kgain = self._create_deterministic_surrogate(det_model, stoch_model, B, Kgain=Kgain)
        self.Kx, self.Kgain = Kx, Kgain
        model_c.setup(dt=1)              # mpc.py:2483 (QUIRK restated: the surrogate always runs with dt = 1)
        super().__init__(model_c, id=id, name=name, plot_backend=plot_backend, stats=stats, use_sx=use_sx,
                         device_index=device_index)
        self._x_ub_p = self._x_lb_p = self._u_ub_p = self._u_lb_p = None
        inf = float('inf')
        self.x_ub_s, self.x_lb_s = [inf] * det_model.n_x, [-inf] * det_model.n_x
        self.u_ub_s, self.u_lb_s = [inf] * det_model.n_u, [-inf] * det_model.n_u
        self._box_constraints_is_set = False

    # ---- the surrogate (mpc.py:2512-2614) ------------------------------------------------------------------------------------
    @staticmethod
    def _discrete_map(m, dt=1.0):
        """Expressions of x+ of the model: the equations of a discrete model; the explicit Runge-Kutta step of a discretised
        one written out (modeling.py:1213-1281; the reference discretises symbolically, so `det_model.ode` IS this map).  The
        step size is the SURROGATE's dt = 1 (mpc.py:2483, :2557), whatever the model was set up with; `Model.linearize` hands
        over the model's own."""
        if not getattr(m, '_symbolic', False):
            raise NotImplementedError("SMPC needs a model written as expressions (set_dynamical_equations); the models of the "
                                      "device zoo carry no expressions to linearise")
        if m._ode is None:
            raise RuntimeError("Model is not set up: no dynamical equations (set_dynamical_equations)")
        if getattr(m, 'n_z', 0):
            raise NotImplementedError("SMPC for models with algebraic states is not built")
        if m._native_discrete or m.erk_order == 0:
            return list(m._ode)
        A, b = _ERK[m.erk_order]
        h = float(dt) / m.n_sub
        xs = list(m.x)
        cur = list(xs)
        for _ in range(m.n_sub):
            ks = []
            for i in range(m.erk_order):
                xi = list(cur)
                for j in range(i):
                    if A[i][j] != 0:
                        xi = [_add(a, _mul(_c(h * A[i][j]), kj)) for a, kj in zip(xi, ks[j])]
                sub = {q: e for q, e in enumerate(xi)}
                ks.append(Expr.substitute(m._ode, lambda n, sub=sub: sub[int(n.value)] if n.op == 'x' else None))
            nxt = list(cur)
            for i in range(m.erk_order):
                if b[i] != 0:
                    nxt = [_add(a, _mul(_c(h * b[i]), ki)) for a, ki in zip(nxt, ks[i])]
            cur = nxt
        return cur

    def _create_deterministic_surrogate(self, det_model, gps, Bw, Kgain=None):
        nx, nu, npar = det_model.n_x, det_model.n_u, det_model.n_p
        gps = [gps] if not isinstance(gps, (list, tuple)) else list(gps)
        if len(gps) > 4:
            raise NotImplementedError("at most 4 learned terms per model")
        Bw = np.atleast_2d(np.asarray(Bw, dtype=float))
        if Bw.shape != (nx, len(gps)):
            if Bw.T.shape == (nx, len(gps)) and nx != len(gps):
                Bw = Bw.T
            else:
                raise ValueError(f"B must have shape ({nx}, {len(gps)}): one column per learned term")
        if getattr(det_model, '_gps', None):
            raise NotImplementedError("the deterministic part of an SMPC model must not contain learned terms itself")
        fmap = self._discrete_map(det_model)
        mc = Model(name=f"{det_model.name or 'model'}_smpc", discrete=True)
        knames = [f'kx_{k}' for k in range(nx * nx)]                     # `ca.SX.sym('kx', n, n)` reshaped column-major (:2578-2581)
        x = mc.set_dynamical_states(list(det_model.dynamical_state_names) + knames)
        u = mc.set_inputs(list(det_model.input_names))
        pnames = list(det_model.parameter_names)
        if Kgain is None:                                                # :2522-2524: the gain becomes n_u n_x parameters
            pnames += [f'kgain_{i}' for i in range(nx * nu)]
        p = mc.set_parameters(pnames)
        if Kgain is None:
            K = [[p[npar + j * nu + i] for j in range(nx)] for i in range(nu)]          # ca.reshape: column-major
        else:
            K = [[_c(Kgain[i, j]) for j in range(nx)] for i in range(nu)]
        Kx = [[x[nx + j * nx + i] for j in range(nx)] for i in range(nx)]
        # learned terms: posterior mean, its Jacobian with respect to (x, u) and the posterior variance at the MEAN state
        mu_d, jgp, kd0 = [], [], []
        for k, gp in enumerate(gps):
            if getattr(gp, '_handle', None) is None:
                raise RuntimeError("The GP has not been set up (trained) yet. Run GaussianProcess.setup() first.")
            feats = list(getattr(gp, 'features', []))
            for f in feats:
                if f not in det_model.dynamical_state_names:            # :2531 `dynamical_state_names.index(i)`
                    raise ValueError(f"'{f}' is not in list")
            if gp.X_train.shape[1] > 256:
                raise NotImplementedError("the posterior variance inside a compiled model is built for up to 256 training points "
                                          "(csrc/hilo_models.h::GP_VAR_MAX)")
            from .gp import is_plain_se
            kern = getattr(gp, 'kernel', None)
            if kern is not None and not is_plain_se(kern.program(len(feats))):
                raise NotImplementedError("the stochastic NMPC's surrogate (posterior mean, its Jacobian and the posterior variance "
                                          "inside the compiled model) is built for squared-exponential kernels")
            fx = [x[det_model.dynamical_state_names.index(f)] for f in feats]
            mean = Expr('gp', fx, value=k, name=(list(getattr(gp, 'labels', [])) or ['gp'])[0])
            mu_d.append(mean)
            row = [_c(0.0)] * (nx + nu)
            for j, f in enumerate(feats):
                row[det_model.dynamical_state_names.index(f)] = Expr('gpd', fx, value=(k, j), name=mean.name)
            jgp.append(row)
            kd0.append(Expr('gpvar', fx, value=k, name=mean.name))
        ng = len(gps)
        ode = [_add(fmap[i], self._dot(Bw[i], mu_d)) for i in range(nx)]               # mean+ = f(mean, u) + B_w mu_d (:2557)
        # Jacobian of the known part (the symbols of the surrogate ARE those of the model: same kinds and positions)
        jode = jacobian(fmap, [x[i] for i in range(nx)] + [u[i] for i in range(nu)])
        Kxu = _matmul(Kx, _T(K))
        Ku = _matmul(_matmul(K, Kx), _T(K))
        Kz = _vcat(_hcat(Kx, Kxu), _hcat(_T(Kxu), Ku))
        Kd0 = [[kd0[a] if a == b else _c(0.0) for b in range(ng)] for a in range(ng)]
        Kd = _madd(Kd0, _matmul(_matmul(jgp, Kz), _T(jgp)))
        Kzd = _matmul(Kz, _T(jgp))
        bigK = _vcat(_hcat(Kz, Kzd), _hcat(_T(Kzd), Kd))
        jodeBw = _hcat(jode, [[_c(Bw[i, k]) for k in range(ng)] for i in range(nx)])
        ode_c = _matmul(jodeBw, _matmul(bigK, _T(jodeBw)))
        ode += [ode_c[i][j] for j in range(nx) for i in range(nx)]       # ca.reshape(ode_c, n^2, 1): column-major
        mc.set_dynamical_equations(ode)
        mc._gps = list(gps)
        return mc, Kx, K

    @staticmethod
    def _dot(row, exprs):
        acc = _c(0.0)
        for c, e in zip(row, exprs):
            acc = _add(acc, _mul(_c(c), e))
        return acc

    # ---- chance constraints (mpc.py:2623-2645) -------------------------------------------------------------------------------
    def _get_chance_constraints(self):
        if self._nlp_options.get('chance_constraints') != 'prs' or not self._box_constraints_is_set:
            return
        n = self._n_x_s
        x = self._model.x
        cu = [math.sqrt(2.0) * _erfinv(2.0 * pr - 1.0) for pr in self._x_ub_p]
        cl = [math.sqrt(2.0) * _erfinv(2.0 * pr - 1.0) for pr in self._x_lb_p]
        sd = [sqrt(self.Kx[i][i] + 1e-8) for i in range(n)]
        rows = [x[i] + cu[i] * sd[i] for i in range(n)] + [-x[i] + cl[i] * sd[i] for i in range(n)]
        ub = [float(v) for v in self.x_ub_s] + [-float(v) for v in self.x_lb_s]
        lb = [-float('inf')] * (2 * n)
        self.stage_constraint.constraint, self.stage_constraint.ub, self.stage_constraint.lb = rows, ub, lb
        self.terminal_constraint.constraint, self.terminal_constraint.ub, self.terminal_constraint.lb = rows, ub, lb

    @staticmethod
    def _sanity_check_probability_values(var, type):
        if var is None:
            return var
        var = [float(v) for v in _wrap_list(var)]
        for i in var:
            if i < 0 or i > 1:
                raise TypeError(f"The probabilities must be between 0 and 1. The variable time {type} has some values"
                                f" ouside this range.")
        return var

    def set_box_constraints(self, *args, **kwargs):
        raise TypeError("set_box_constraints is not available in stochastic MPC. Use 'set_box_chance_constraints' instead.")

    def set_box_chance_constraints(self, x_ub=None, x_lb=None, u_ub=None, u_lb=None, y_ub=None, y_lb=None, z_ub=None, z_lb=None,
                                   *args, **kwargs):
        """mpc.py:2667-2745: bounds on the n_x original states plus the probabilities `x_ub_p` / `x_lb_p` (default 0.954) they have
        to hold with; the covariance states get [0, inf) on the diagonal and no bounds off it."""
        n = self._n_x_s
        inf = float('inf')
        if y_ub is not None or y_lb is not None or z_ub is not None or z_lb is not None:
            raise NotImplementedError("chance constraints on measurements / algebraic states are not offloaded")
        if x_ub is not None:
            x_ub = [float(v) for v in _wrap_list(x_ub)]
            self.x_ub_s = list(x_ub)
            x_ub = x_ub + [inf] * (n * n)
        self._x_ub_p = self._sanity_check_probability_values(kwargs.get('x_ub_p', np.ones(n) * 0.954), 'x_ub')
        if x_lb is not None:
            x_lb = [float(v) for v in _wrap_list(x_lb)]
            self.x_lb_s = list(x_lb)
            x_lb = x_lb + [0.0 if i == j else -inf for i in range(n) for j in range(n)]
        self._x_lb_p = self._sanity_check_probability_values(kwargs.get('x_lb_p', np.ones(n) * 0.954), 'x_lb')
        for pr, what in ((self._x_ub_p, 'x_ub_p'), (self._x_lb_p, 'x_lb_p')):
            if len(pr) != n:
                raise ValueError(f"{what} must have {n} entries")
        super().set_box_constraints(x_ub=x_ub, x_lb=x_lb, u_ub=u_ub, u_lb=u_lb)
        self._box_constraints_is_set = True

    def set_custom_constraints_function(self, *args, **kwargs):
        raise NotImplementedError("set_custom_constraints_function is not yet implemented for SMPC class. ")

    def set_stage_constraints(self, *args, **kwargs):
        raise NotImplementedError("set_stage_constraints is not yet implemented for SMPC class. ")

    def set_terminal_constraints(self, *args, **kwargs):
        raise NotImplementedError("set_terminal_constraints is not yet implemented for SMPC class. ")

    def set_nlp_options(self, *args, **kwargs):
        """optimizer.py:1407-1421: the NMPC options plus `chance_constraints` (only 'prs', the default)."""
        given = dict(args[0]) if (args and isinstance(args[0], dict)) else dict(kwargs)
        cc = given.pop('chance_constraints', 'prs')
        if cc not in ('prs',):
            raise ValueError(f"The option chance_constraints is set to value {cc} but the only allowed values are ['prs'].")
        super().set_nlp_options(given)
        self._nlp_options['chance_constraints'] = cc

    # ---- setup / optimize (mpc.py:2760-2798) ---------------------------------------------------------------------------------
    def _weights(self):
        """`quad_stage_cost.Q` / `.R` (modeling.py:492-512): half the Hessian of the quadratic stage cost."""
        nxa, nu = self._n_x, self._n_u
        Q, R = np.zeros((nxa, nxa)), np.zeros((nu, nu))
        for kind, ind, W, _ in self.quad_stage_cost._terms:
            if kind == 'states':
                Q[np.ix_(ind, ind)] += W
            else:                                                        # inputs and input changes are both quadratic in u
                R[np.ix_(ind, ind)] += W
        return Q, R

    def setup(self, options=None, solver_options=None):
        self.set_nlp_options(options or {})
        self._get_chance_constraints()
        n, nu = self._n_x_s, self._n_u_s
        Q, R = self._weights()
        Ku = _matmul(_matmul(self.Kgain, self.Kx), _T(self.Kgain))
        cost = _c(0.0)                                                   # trace(Q Kx) + trace(R Ku), mpc.py:2766-2771
        for i in range(n):
            for j in range(n):
                cost = _add(cost, _mul(_c(Q[i, j]), self.Kx[j][i]))
        for i in range(nu):
            for j in range(nu):
                cost = _add(cost, _mul(_c(R[i, j]), Ku[j][i]))
        self.stage_cost.cost = cost
        cc = self._nlp_options['chance_constraints']
        super().setup(options=None, solver_options=solver_options)
        self._nlp_options['chance_constraints'] = cc

    def optimize(self, x0, cp=None, tvp=None, v0=None, runs=0, fix_x0=True, **kwargs):
        """`x0`: mean of the state, [n_x] or [B, n_x]; `cov_x0`: its covariance, [n_x, n_x] or [B, n_x, n_x]; `Kgain`: gain of the
        ancillary controller, [n_u, n_x] or [B, n_u, n_x], unless it was given to the constructor."""
        cov_x0 = kwargs.pop('cov_x0', None)
        if cov_x0 is None:
            raise ValueError("To solve the SMPC you need to provide an intial condition for state covariance values. "
                             "Please pass a 'cov_x0' as well.")
        n, nu = self._n_x_s, self._n_u_s
        Kgain = kwargs.pop('Kgain', None)
        if not self._nlp_setup_done:
            raise ValueError("Howdy! You need to setup the MPC before optimizing. Run .setup() on the MPC object.")
        on_dev = isinstance(x0, torch.Tensor)

        def arr(a):
            """Everything in the family of x0: device tensors stay on the device, host data stays numpy."""
            if on_dev:
                return to_dev(a, self._dev).to(torch.float64)
            return np.asarray(a.cpu().numpy() if isinstance(a, torch.Tensor) else a, dtype=float)

        cat = (lambda parts: torch.cat(parts, dim=1)) if on_dev else (lambda parts: np.concatenate(parts, axis=1))
        tile = (lambda a, r: a.expand(r, -1)) if on_dev else (lambda a, r: np.broadcast_to(a, (r, a.shape[1])))
        swap = (lambda a: a.transpose(1, 2)) if on_dev else (lambda a: np.swapaxes(a, 1, 2))
        x = arr(x0)
        single = x.ndim <= 1 or (x.ndim == 2 and x.shape[1] == 1 and x.shape[0] == n and n != 1)
        x = x.reshape(-1, n)
        Bn = x.shape[0]
        c = arr(cov_x0).reshape(-1, n, n)
        if c.shape[0] == 1 and Bn > 1:
            c = c.expand(Bn, n, n) if on_dev else np.broadcast_to(c, (Bn, n, n))
        if c.shape[0] != Bn:
            raise ValueError(f"cov_x0 holds {c.shape[0]} matrices for {Bn} initial states")
        xa = cat([x, swap(c).reshape(Bn, n * n)])                            # ca.reshape(cov_x0, n^2, 1): column-major
        if not self._Kgain_is_set:
            if Kgain is None:
                raise ValueError("It looks like you have not passed the gain of the ancillary controller yet. "
                                 "Please provide a 'Kgain' to the optimize method.")
            k = arr(Kgain)
            if k.ndim == 0 or k.shape[-1] * (k.shape[-2] if k.ndim > 1 else 1) != nu * n:
                if (k.numel() if on_dev else k.size) == 1:                   # tests/test_SMPC.py: `Kgain=0` for a 1 x 1 gain
                    k = k.reshape(1, 1, 1) * (torch.ones(1, nu, n, dtype=torch.float64, device=self._dev) if on_dev
                                             else np.ones((1, nu, n)))
                else:
                    raise ValueError(f"Kgain must have shape ({nu}, {n})")
            k = k.reshape(-1, nu, n)
            kg = swap(k).reshape(k.shape[0], nu * n)                         # ca.reshape(Kgain, n_x n_u, 1): column-major
            if cp is not None:
                cpt = arr(cp).reshape(-1, self._n_p_s)
                rows = max(cpt.shape[0], kg.shape[0])
                cp = cat([tile(cpt, rows), tile(kg, rows)])
            else:
                cp = kg
            cp = cp.contiguous() if on_dev else np.ascontiguousarray(cp)
        xa = xa.contiguous() if on_dev else np.ascontiguousarray(xa)
        if single:
            xa = xa[0]
            cp = cp[0] if (cp is not None and cp.shape[0] == 1 and cp.ndim == 2) else cp
        return super().optimize(xa, cp=cp, tvp=tvp, v0=v0, runs=runs, fix_x0=fix_x0, **kwargs)
