"""Gaussian-process regression (exact inference) on the GPU.

API mirror of `hilo_mpc/modules/machine_learning/gp/` for the inference path:
`Kernel.<factory>(...)` / `Mean.<factory>(...)` with `+`, `*`, `**` composition (kernel.py:207-435, 1426-1666;
mean.py:624-766), `Kernel.__call__(X, X_bar)` -> covariance matrix, `GaussianProcess(features, labels, ...)`,
`set_training_data`, `setup`, `predict(X_query, noise_free)`, `log_marginal_likelihood()`.
Inputs are feature-major (n_features x n_observations) exactly like the reference (kernel.py:97-140).

A kernel/mean tree is compiled to a flat postfix program (opcodes in include/hilo_hip.h) that the device
interprets; the numeric work (covariance matrices, Cholesky, alpha, L^-1, predictions) is done by
libhilo_hip.so only.  Hyper-parameter fitting (`fit_model`) drives the device log marginal likelihood
from a host quasi-Newton loop; an analytic device gradient is the next step.
"""
import ctypes as C
from math import factorial, gamma as gamma_fun

import numpy as np
import torch

from . import _lib
from ._device import device, to_dev, ptr, stream_ptr

# opcodes (include/hilo_hip.h)
K_CONST, K_GAMMAEXP, K_MATERN, K_RQ, K_PP, K_POLY, K_NN, K_PERIODIC = range(8)
K_SUM, K_PRODUCT, K_POWER, K_XX_BEGIN = 16, 17, 18, 19
M_CONST, M_POLY, M_SUM, M_PRODUCT, M_POWER, M_SCALE = 32, 33, 48, 49, 50, 51


def _is_list_like(v):
    return isinstance(v, (list, tuple, np.ndarray))


def _node(op, active, params):
    return [float(op), float(len(active))] + [float(a) for a in active] + [float(len(params))] + \
           [float(p) for p in params]


def _log(v, is_variance=False):
    """kernel.py:127-130: hyper-parameters enter as logs; `*variance*` as log(value)/2."""
    with np.errstate(divide='ignore'):
        lg = np.log(np.asarray(v, dtype=float))
    return lg / 2. if is_variance else lg


def kernel_expr(prog, f, t):
    """k(f, t) of a kernel program as an expression tree (hilo_mpc_amd/expr.py) - the formulas of the device interpreter
    (csrc/hilo_gp.hip::eval_kernel, kernel.py:465-1003, :1562-1627), with the features `f` and the training point `t` as lists
    of expressions.  Used for a learned term inside a run-time compiled model whose kernel is not the plain squared
    exponential (hilo_mpc_amd/model.py::substitute_from): the kernel is compiled into the model source and differentiated by the
    scalar type it is evaluated with.  Distances enter square roots / fractional powers with 1e-300 added, so that the Taylor
    coefficients stay finite where a query coincides with a training point (the value is unchanged)."""
    from . import expr as E
    prog = [float(v) for v in prog]
    pos, st = 0, []
    while pos < len(prog):
        op, na = int(prog[pos]), int(prog[pos + 1])
        act = [int(v) for v in prog[pos + 2:pos + 2 + na]]
        npar = int(prog[pos + 2 + na])
        par = prog[pos + 3 + na:pos + 3 + na + npar]
        pos += 3 + na + npar

        def d2(M):
            s2 = E.Expr.wrap(0.0)
            for k, a in enumerate(act):
                df = f[a] - t[a]
                s2 = s2 + float(M[k]) * (df * df)
            return s2

        if op == K_CONST:
            v = E.Expr.wrap(par[0])
        elif op == K_GAMMAEXP:
            q = d2(par[3:3 + na])
            e = q if par[2] == 1.0 else (q + 1e-300) ** float(par[2])
            v = par[0] * E.exp(-par[1] * e)
        elif op == K_MATERN:
            nc = int(par[2])
            d = E.sqrt((par[1] * par[1]) * d2(par[3 + nc:3 + nc + na]) + 1e-300)
            poly = 1.0 + d * par[3]
            for k in range(1, nc):
                poly = 1.0 + d * par[3 + k] * poly
            v = par[0] * E.exp(-d) * poly
        elif op == K_RQ:
            v = par[0] * (1.0 + (0.5 / par[1]) * d2(par[2:2 + na])) ** float(-par[1])
        elif op == K_SUM:
            b, a = st.pop(), st.pop()
            v = a + b
        elif op == K_PRODUCT:
            b, a = st.pop(), st.pop()
            v = a * b
        else:
            raise NotImplementedError("a learned term inside a run-time compiled model takes squared-exponential / gamma-exponential, "
                                      "Matern, rational-quadratic and constant kernels and their sums and products")
        st.append(v)
    if len(st) != 1:
        raise ValueError("malformed kernel program")
    return st[0]


def is_plain_se(prog):
    """The kernel program is ONE squared-exponential node (the test of csrc/hilo_gp.hip::gp_pack_se)."""
    prog = [float(v) for v in prog]
    na = int(prog[1])
    return int(prog[0]) == K_GAMMAEXP and 1 <= na <= 8 and len(prog) == 3 + na + 3 + na and int(prog[2 + na]) == 3 + na and \
        prog[3 + na + 1] == 0.5 and prog[3 + na + 2] == 1.0


# =================================================================================================
# Kernels
# =================================================================================================
class Kernel:
    """Base class + factory namespace (kernel.py:49-435)."""
    acronym = None

    def __init__(self, active_dims=None):
        self.active_dims = None if active_dims is None else [int(a) for a in np.atleast_1d(active_dims)]
        if not hasattr(self, '_bounds'):
            self._bounds = {}

    def __init_subclass__(cls, **kw):
        # every kernel constructor takes `bounds=`: record it after the constructor ran (the outermost class's wins)
        super().__init_subclass__(**kw)
        orig = cls.__dict__.get('__init__')
        if orig is None:
            return
        import functools
        import inspect
        sig = inspect.signature(orig)
        if 'bounds' not in sig.parameters:
            return

        @functools.wraps(orig)
        def init(self, *a, **k):
            orig(self, *a, **k)
            self._set_bounds(sig.bind(self, *a, **k).arguments.get('bounds'))
        cls.__init__ = init

    def _set_bounds(self, bounds):
        """`bounds={'signal_variance': 'fixed', 'length_scales': (1e-2, 1e2)}` (util/machine_learning.py:283-334): 'fixed'
        keeps the hyper-parameter out of `fit_model`, a number is a lower bound, a pair a box (on the value)."""
        self._bounds = {}
        for k, b in dict(bounds or {}).items():
            if k not in self._hyper:
                raise KeyError(f"'{k}' is not a hyper-parameter of the {self.acronym} kernel ({list(self._hyper)})")
            if isinstance(b, str):
                if b != 'fixed':
                    raise ValueError(f"Unsupported bounds '{b}'")
                self._bounds[k] = 'fixed'
            elif isinstance(b, (int, float)):
                self._bounds[k] = (float(b), np.inf)
            else:
                b = tuple(float(v) for v in b)
                if len(b) == 1:
                    b = (b[0], np.inf)
                if len(b) != 2 or not b[0] < b[1]:
                    raise ValueError("Lower bound not smaller than upper bound!")
                self._bounds[k] = b

    def hyperparameter_bounds(self):
        """One entry per scalar hyper-parameter, parallel to `hyperparameter_handles()`: 'fixed', (lb, ub) or None."""
        return [self._bounds.get(a) for _, a, _ in Kernel.hyperparameter_handles(self)]

    # ---- composition (kernel.py:276-296) ----
    def __add__(self, other):
        return Sum(self, other)

    def __mul__(self, other):
        return Product(self, other)

    def __pow__(self, power, modulo=None):
        return Power(self, power)

    def __radd__(self, other):
        return Sum(other, self)

    def __rmul__(self, other):
        return Product(other, self)

    # ---- hyper-parameters (kernel.py:300-350: `hyperparameters`, in declaration order) ----
    _hyper = ()     # attribute names, in the reference's order

    def hyperparameter_handles(self):
        """[(owner, attribute, index or None)]: one entry per scalar hyper-parameter (a vector of ARD length scales expands)."""
        out = []
        for a in self._hyper:
            v = getattr(self, a)
            if _is_list_like(v):
                out += [(self, a, i) for i in range(len(v))]
            else:
                out.append((self, a, None))
        return out

    @property
    def hyperparameter_names(self):
        return [f"{self.acronym}.{a}" + ('' if i is None else f"_{i}") for _, a, i in self.hyperparameter_handles()]

    # ---- evaluation ----
    def _active(self, nf):
        return list(range(nf)) if self.active_dims is None else self.active_dims

    def program(self, nf):
        raise NotImplementedError

    def __call__(self, X, X_bar=None):
        """Covariance matrix K[i, j] = k(X[:, i], X_bar[:, j]) (kernel.py:97-140)."""
        host = not isinstance(X, torch.Tensor)
        if X_bar is not None and isinstance(X_bar, torch.Tensor) == host:
            raise ValueError("X and X_bar need to have the same type")
        dev = device()
        Xd = to_dev(np.atleast_2d(X) if host else X, dev)
        Xd = Xd.reshape(1, -1) if Xd.ndim == 1 else Xd
        Xb = Xd if X_bar is None else to_dev(np.atleast_2d(X_bar) if host else X_bar, dev)
        Xb = Xb.reshape(1, -1) if Xb.ndim == 1 else Xb
        assert Xd.shape[0] == Xb.shape[0], "X and X_bar do not have the same input space dimensions"
        nf = Xd.shape[0]
        prog = np.asarray(self.program(nf), dtype=np.float64)
        K = torch.empty(Xd.shape[1], Xb.shape[1], dtype=torch.float64, device=dev)
        _lib.check(_lib.lib().hilo_gp_kernel_matrix(dev.index, nf, prog.ctypes.data, prog.size, Xd.shape[1], ptr(Xd),
                                                    Xb.shape[1], ptr(Xb), ptr(K), stream_ptr(dev)))
        return K.cpu().numpy() if host else K

    # ---- factories (kernel.py:228-435) ----
    @staticmethod
    def constant(bias=1., bounds=None):
        return ConstantKernel(bias=bias, bounds=bounds)

    @staticmethod
    def squared_exponential(active_dims=None, signal_variance=1., length_scales=1., ard=False, bounds=None):
        return SquaredExponentialKernel(active_dims, signal_variance, length_scales, ard, bounds=bounds)

    @staticmethod
    def exponential(active_dims=None, signal_variance=1., length_scales=1., ard=False, bounds=None):
        return ExponentialKernel(active_dims, signal_variance, length_scales, ard, bounds=bounds)

    @staticmethod
    def matern_32(active_dims=None, signal_variance=1., length_scales=1., ard=False, bounds=None):
        return Matern32Kernel(active_dims, signal_variance, length_scales, ard, bounds=bounds)

    @staticmethod
    def matern_52(active_dims=None, signal_variance=1., length_scales=1., ard=False, bounds=None):
        return Matern52Kernel(active_dims, signal_variance, length_scales, ard, bounds=bounds)

    @staticmethod
    def rational_quadratic(active_dims=None, signal_variance=1., length_scales=1., alpha=1., ard=False, bounds=None):
        return RationalQuadraticKernel(active_dims, signal_variance, length_scales, alpha, ard, bounds=bounds)

    @staticmethod
    def piecewise_polynomial(degree, active_dims=None, signal_variance=1., length_scales=1., ard=False, bounds=None):
        return PiecewisePolynomialKernel(degree, active_dims, signal_variance, length_scales, ard, bounds=bounds)

    @staticmethod
    def polynomial(degree, active_dims=None, signal_variance=1., offset=1., bounds=None):
        return PolynomialKernel(degree, active_dims, signal_variance, offset, bounds=bounds)

    @staticmethod
    def linear(active_dims=None, signal_variance=1., bounds=None):
        return LinearKernel(active_dims, signal_variance, bounds=bounds)

    @staticmethod
    def neural_network(active_dims=None, signal_variance=1., weight_variance=1., bounds=None):
        return NeuralNetworkKernel(active_dims, signal_variance, weight_variance, bounds=bounds)

    @staticmethod
    def periodic(active_dims=None, signal_variance=1., length_scales=1., period=1., bounds=None):
        return PeriodicKernel(active_dims, signal_variance, length_scales, period, bounds=bounds)


class ConstantKernel(Kernel):
    """kernel.py:436-485: exp(2 log bias)."""
    acronym = "Const"
    _hyper = ('bias',)

    def __init__(self, bias=1., bounds=None):
        super().__init__()
        self.bias = bias

    def program(self, nf):
        return _node(K_CONST, [], [np.exp(2 * _log(self.bias))])


class StationaryKernel(Kernel):
    """kernel.py:488-562."""
    acronym = "Stat"
    _hyper = ('length_scales', 'signal_variance')

    def __init__(self, active_dims=None, length_scales=1., ard=False, bounds=None):
        super().__init__(active_dims=active_dims)
        if ard and active_dims is None:
            raise ValueError("The key word 'ard' can only be set to True if the key word 'active_dims' was supplied")
        if active_dims is not None and _is_list_like(length_scales):
            if len(active_dims) != len(length_scales):
                raise ValueError(f"Dimension mismatch between 'active_dims' ({len(active_dims)}) and the number of "
                                 f"length_scales ({len(length_scales)})")
        if not _is_list_like(length_scales) and ard:
            length_scales = len(active_dims) * [length_scales]
        self.length_scales = length_scales

    def is_isotropic(self):
        return not _is_list_like(self.length_scales)

    def _M(self, n_active):
        """kernel.py:538-555: diag(exp(-2 log l))."""
        ll = _log(self.length_scales)
        if self.is_isotropic():
            return [float(np.exp(-2 * ll))] * n_active
        if ll.size != n_active:
            raise ValueError("Length scales vector dimension does not equal input space dimension.")
        return list(np.exp(-2 * ll))


class GammaExponentialKernel(StationaryKernel):
    """kernel.py:565-701: exp(2 log s - alpha d2^(p/2)), p = 2/(1 - exp(-log(gamma/(2-gamma))))."""
    acronym = "GE"

    def __init__(self, active_dims=None, signal_variance=1., alpha=None, gamma=1., length_scales=1., ard=False,
                 bounds=None):
        super().__init__(active_dims, length_scales, ard)
        self.signal_variance = signal_variance
        if gamma == 2.:
            raise ValueError("Value of the hyperparameter 'gamma' is set to 2. Use the squared exponential kernel "
                             "instead.")
        self.gamma = gamma / (2 - gamma)
        self.alpha = 1. if alpha is None else alpha

    def _p_alpha(self):
        with np.errstate(divide='ignore'):
            p = 2 / (1 - np.exp(-np.log(self.gamma)))
        return p, self.alpha

    def program(self, nf):
        ad = self._active(nf)
        p, alpha = self._p_alpha()
        sf2 = np.exp(2 * _log(self.signal_variance, True))
        return _node(K_GAMMAEXP, ad, [sf2, alpha, p / 2] + self._M(len(ad)))


class SquaredExponentialKernel(GammaExponentialKernel):
    """kernel.py:704-734 (gamma = 2 -> p = 2, alpha = 1/2)."""
    acronym = "SE"

    def __init__(self, active_dims=None, signal_variance=1., length_scales=1., ard=False, bounds=None):
        StationaryKernel.__init__(self, active_dims, length_scales, ard)
        self.signal_variance = signal_variance
        self.gamma = 2.
        self.alpha = .5

    def _p_alpha(self):
        return 2., .5


class MaternKernel(StationaryKernel):
    """kernel.py:737-826."""
    acronym = "Matern"

    def __init__(self, p, active_dims=None, signal_variance=1., length_scales=1., ard=False, bounds=None):
        super().__init__(active_dims, length_scales, ard)
        self.signal_variance = signal_variance
        self._p = p

    def _poly(self):
        p = self._p
        if p > 1:
            g1, g2 = gamma_fun(p + 1), gamma_fun(2 * p + 1)
            poly = [g1 / g2 * factorial(p + k) / (factorial(k) * factorial(p - k)) * 2. ** (p - k)
                    for k in range(p - 1)]
        else:
            poly = []
        if p == 0:
            poly.append(0.)
        elif p >= 1:
            poly.append(1.)
        for k in range(len(poly) - 1, 0, -1):
            poly[k - 1] /= poly[k]
        return poly

    def program(self, nf):
        ad = self._active(nf)
        poly = self._poly()
        sf2 = np.exp(2 * _log(self.signal_variance, True))
        return _node(K_MATERN, ad, [sf2, np.sqrt(2 * (self._p + .5)), len(poly)] + poly + self._M(len(ad)))


class ExponentialKernel(MaternKernel):
    acronym = "E"

    def __init__(self, active_dims=None, signal_variance=1., length_scales=1., ard=False, bounds=None):
        super().__init__(0, active_dims, signal_variance, length_scales, ard)


class Matern32Kernel(MaternKernel):
    acronym = "M32"

    def __init__(self, active_dims=None, signal_variance=1., length_scales=1., ard=False, bounds=None):
        super().__init__(1, active_dims, signal_variance, length_scales, ard)


class Matern52Kernel(MaternKernel):
    acronym = "M52"

    def __init__(self, active_dims=None, signal_variance=1., length_scales=1., ard=False, bounds=None):
        super().__init__(2, active_dims, signal_variance, length_scales, ard)


class RationalQuadraticKernel(StationaryKernel):
    """kernel.py:919-1003."""
    acronym = "RQ"
    _hyper = ('length_scales', 'signal_variance', 'alpha')

    def __init__(self, active_dims=None, signal_variance=1., length_scales=1., alpha=1., ard=False, bounds=None):
        super().__init__(active_dims, length_scales, ard)
        self.signal_variance = signal_variance
        self.alpha = alpha

    def program(self, nf):
        ad = self._active(nf)
        sf2 = np.exp(2 * _log(self.signal_variance, True))
        return _node(K_RQ, ad, [sf2, np.exp(_log(self.alpha))] + self._M(len(ad)))


class PiecewisePolynomialKernel(StationaryKernel):
    """kernel.py:1006-1109."""
    acronym = "PP"

    def __init__(self, degree, active_dims=None, signal_variance=1., length_scales=1., ard=False, bounds=None):
        super().__init__(active_dims, length_scales, ard)
        self.signal_variance = signal_variance
        self.degree = degree

    @property
    def degree(self):
        return self._q

    @degree.setter
    def degree(self, value):
        if value not in [0, 1, 2, 3]:
            raise ValueError("The property 'degree' has to be one of the following integers: 0, 1, 2, 3")
        self._q = value

    def program(self, nf):
        ad = self._active(nf)
        j = np.floor(len(ad) / 2) + self._q + 1
        sf2 = np.exp(2 * _log(self.signal_variance, True))
        return _node(K_PP, ad, [sf2, self._q, j] + self._M(len(ad)))


class DotProductKernel(Kernel):
    acronym = "Dot"

    def __init__(self, active_dims=None, signal_variance=1., offset=1., bounds=None):
        super().__init__(active_dims=active_dims)
        self.signal_variance = signal_variance
        self.offset = offset


class PolynomialKernel(DotProductKernel):
    """kernel.py:1160-1234."""
    acronym = "Poly"
    _hyper = ('signal_variance', 'offset')

    def __init__(self, degree, active_dims=None, signal_variance=1., offset=1., bounds=None):
        super().__init__(active_dims, signal_variance, offset)
        self.degree = degree

    def program(self, nf):
        sf2 = np.exp(2 * _log(self.signal_variance, True))
        return _node(K_POLY, self._active(nf), [sf2, np.exp(_log(self.offset)), self.degree])


class LinearKernel(PolynomialKernel):
    """kernel.py:1237-1259 (degree 1, offset 0)."""
    acronym = "Lin"
    _hyper = ('signal_variance',)

    def __init__(self, active_dims=None, signal_variance=1., bounds=None):
        super().__init__(1, active_dims, signal_variance)
        self.offset = 0.


class NeuralNetworkKernel(Kernel):
    """kernel.py:1262-1332."""
    acronym = "NN"
    _hyper = ('signal_variance', 'weight_variance')

    def __init__(self, active_dims=None, signal_variance=1., weight_variance=1., bounds=None):
        super().__init__(active_dims=active_dims)
        self.signal_variance = signal_variance
        self.weight_variance = weight_variance

    def program(self, nf):
        return _node(K_NN, self._active(nf), [np.exp(2 * _log(self.signal_variance, True)),
                                             np.exp(2 * _log(self.weight_variance, True))])


class PeriodicKernel(Kernel):
    """kernel.py:1335-1423 (scalar expression only for one active dimension)."""
    acronym = "Periodic"
    _hyper = ('signal_variance', 'length_scales', 'period')

    def __init__(self, active_dims=None, signal_variance=1., length_scales=1., period=1., bounds=None):
        super().__init__(active_dims=active_dims)
        self.signal_variance = signal_variance
        self.length_scales = length_scales
        self.period = period

    def program(self, nf):
        ad = self._active(nf)
        if len(ad) != 1:
            raise ValueError("The periodic covariance function is only defined for one active dimension")
        return _node(K_PERIODIC, ad, [2 * _log(self.signal_variance, True), np.exp(_log(self.length_scales)),
                                      np.exp(_log(self.period))])


class KernelOperator(Kernel):
    def __init__(self, kernel_1, kernel_2=None):
        super().__init__()
        self.kernel_1 = kernel_1
        self.kernel_2 = kernel_2

    def hyperparameter_handles(self):
        out = self.kernel_1.hyperparameter_handles() if isinstance(self.kernel_1, Kernel) else []
        if isinstance(self.kernel_2, Kernel):
            out += self.kernel_2.hyperparameter_handles()
        return out

    def hyperparameter_bounds(self):
        out = self.kernel_1.hyperparameter_bounds() if isinstance(self.kernel_1, Kernel) else []
        return out + (self.kernel_2.hyperparameter_bounds() if isinstance(self.kernel_2, Kernel) else [])

    @property
    def hyperparameter_names(self):
        out = self.kernel_1.hyperparameter_names if isinstance(self.kernel_1, Kernel) else []
        return out + (self.kernel_2.hyperparameter_names if isinstance(self.kernel_2, Kernel) else [])


class Sum(KernelOperator):
    def program(self, nf):
        return self.kernel_1.program(nf) + self.kernel_2.program(nf) + _node(K_SUM, [], [])


class Product(KernelOperator):
    def program(self, nf):
        return self.kernel_1.program(nf) + self.kernel_2.program(nf) + _node(K_PRODUCT, [], [])


class Power(KernelOperator):
    """kernel.py:1630-1666 - evaluates its child at (x, x) (`self.kernel_1(x)`, :1651)."""

    def __init__(self, kernel, power):
        super().__init__(kernel)
        self.power = power

    def program(self, nf):
        return _node(K_XX_BEGIN, [], []) + self.kernel_1.program(nf) + _node(K_POWER, [], [self.power])


# =================================================================================================
# Means (mean.py)
# =================================================================================================
class Mean:
    acronym = None

    def __init__(self, active_dims=None):
        self.active_dims = None if active_dims is None else [int(a) for a in np.atleast_1d(active_dims)]

    def __add__(self, other):
        return MeanSum(self, other)

    def __mul__(self, other):
        if isinstance(other, Mean):
            return MeanProduct(self, other)
        return MeanScale(self, other)

    __rmul__ = __mul__

    def __pow__(self, power, modulo=None):
        return MeanPower(self, power)

    def program(self, nf):
        raise NotImplementedError

    def trainable_hyperparameters(self):
        """Names of the mean's hyper-parameters that `fit_model` optimises together with the kernel's (mean.py: every
        `Hyperparameter` that is not `fixed`; gp.py:408-414)."""
        return [f"{o.acronym}.{a}" + ('' if i is None else f"_{i}") for o, a, i in self.hyperparameter_handles()]

    def hyperparameter_handles(self):
        """[(object, attribute, index or None)] of the trainable (non-fixed) hyper-parameters; their optimisation variable is
        the value itself (`positive=False` in the reference, mean.py:278)."""
        return []

    def __call__(self, X):
        """mean.py:90-116: returns (1 x n_obs)."""
        host = not isinstance(X, torch.Tensor)
        dev = device()
        Xd = to_dev(np.atleast_2d(X) if host else X, dev)
        nf, n = Xd.shape
        prog = np.asarray(self.program(nf), dtype=np.float64)
        mu = torch.empty(1, n, dtype=torch.float64, device=dev)
        _lib.check(_lib.lib().hilo_gp_mean(dev.index, nf, prog.ctypes.data, prog.size, n, ptr(Xd), ptr(mu),
                                           stream_ptr(dev)))
        return mu.cpu().numpy() if host else mu

    @staticmethod
    def constant(bias=1., hyperprior=None, **kwargs):
        return ConstantMean(bias, hyperprior, **kwargs)

    @staticmethod
    def zero():
        return ZeroMean()

    @staticmethod
    def one():
        return OneMean()

    @staticmethod
    def polynomial(degree, active_dims=None, coefficient=1., offset=1., hyperprior=None, **kwargs):
        return PolynomialMean(degree, active_dims, coefficient, offset, hyperprior, **kwargs)

    @staticmethod
    def linear(active_dims=None, coefficient=1., hyperprior=None, **kwargs):
        return LinearMean(active_dims, coefficient, hyperprior, **kwargs)


class ConstantMean(Mean):
    acronym = "Const"

    def __init__(self, bias=1., hyperprior=None, **kwargs):
        super().__init__()
        self.bias = bias
        self._fixed = {k for k, v in (kwargs.get('bounds') or {}).items() if v == 'fixed'}     # mean.py:270-277

    def hyperparameter_handles(self):
        return [] if 'bias' in self._fixed or type(self) is not ConstantMean else [(self, 'bias', None)]

    def program(self, nf):
        return _node(M_CONST, [], [self.bias])


class ZeroMean(ConstantMean):
    acronym = "Zero"

    def __init__(self):
        super().__init__(0.)


class OneMean(ConstantMean):
    acronym = "One"

    def __init__(self):
        super().__init__(1.)


class PolynomialMean(Mean):
    """mean.py:329-470: (M^T x[active] + offset)^p."""
    acronym = "Poly"

    def __init__(self, degree, active_dims=None, coefficient=1., offset=1., hyperprior=None, **kwargs):
        super().__init__(active_dims)
        if active_dims is not None and _is_list_like(coefficient):
            if len(active_dims) != len(coefficient):
                raise ValueError(f"Dimension mismatch between 'active_dims' ({len(active_dims)}) and the number of "
                                 f"coefficients ({len(coefficient)})")
        self.coefficient = coefficient
        self.offset = offset
        self.degree = degree
        self._fixed = {k for k, v in (kwargs.get('bounds') or {}).items() if v == 'fixed'}     # mean.py:385-405

    def hyperparameter_handles(self):
        out = []
        if 'coefficient' not in self._fixed:
            out += [(self, 'coefficient', i) for i in range(len(self.coefficient))] if _is_list_like(self.coefficient) else \
                [(self, 'coefficient', None)]
        if type(self) is PolynomialMean and 'offset' not in self._fixed:                       # the linear mean's offset is 0, fixed
            out.append((self, 'offset', None))
        return out

    def program(self, nf):
        ad = list(range(nf)) if self.active_dims is None else self.active_dims
        c = np.asarray(self.coefficient, dtype=float)
        if c.ndim == 0:
            c = c * np.ones(len(ad))
        if c.size != len(ad):
            raise ValueError("Coefficient vector dimension does not equal input space dimension.")
        return _node(M_POLY, ad, [self.offset, self.degree] + list(c))


class LinearMean(PolynomialMean):
    acronym = "Lin"
    _hyper = ('signal_variance',)

    def __init__(self, active_dims=None, coefficient=1., hyperprior=None, **kwargs):
        super().__init__(1, active_dims, coefficient, hyperprior, **kwargs)
        self.offset = 0.


class _MeanOp(Mean):
    def __init__(self, mean_1, mean_2=None):
        super().__init__()
        self.mean_1, self.mean_2 = mean_1, mean_2

    def hyperparameter_handles(self):
        return [h for m in (self.mean_1, self.mean_2) if isinstance(m, Mean) for h in m.hyperparameter_handles()]


class MeanSum(_MeanOp):
    def program(self, nf):
        return self.mean_1.program(nf) + self.mean_2.program(nf) + _node(M_SUM, [], [])


class MeanProduct(_MeanOp):
    def program(self, nf):
        return self.mean_1.program(nf) + self.mean_2.program(nf) + _node(M_PRODUCT, [], [])


class MeanPower(_MeanOp):
    def __init__(self, mean, power):
        super().__init__(mean)
        self.power = power

    def program(self, nf):
        return self.mean_1.program(nf) + _node(M_POWER, [], [self.power])


class MeanScale(_MeanOp):
    def __init__(self, mean, scale):
        super().__init__(mean)
        self.scale = scale

    def program(self, nf):
        return self.mean_1.program(nf) + _node(M_SCALE, [], [self.scale])


# =================================================================================================
# GaussianProcess (gp.py:112-236, 522-641, 699-718)
# =================================================================================================
class GaussianProcess:
    def __init__(self, features, labels, inference=None, likelihood=None, mean=None, kernel=None, noise_variance=1.,
                 hyperprior=None, id=None, name=None, solver=None, solver_options=None, **kwargs):
        if not _is_list_like(features):
            features = [features]
        if not _is_list_like(labels):
            labels = [labels]
        if len(labels) > 1:
            raise ValueError("Training a GP on multiple labels is not supported. Please use 'MultiOutputGP' to train "
                             "GPs on multiple labels.")
        if inference is not None and (not isinstance(inference, str) or inference.replace(' ', '_').lower() != 'exact'):
            raise ValueError(f"Inference '{inference}' not recognized")       # only exact inference exists (inference.py)
        if likelihood is not None and (not isinstance(likelihood, str) or
                                       likelihood.replace("'", "").replace(' ', '_').lower() != 'gaussian'):
            raise ValueError("Exact inference is only applicable with Gaussian likelihood. Choose a different "
                             "inference method in order to use other likelihoods.")
        self.features, self.labels = list(features), list(labels)
        self.name = name
        self.mean = Mean.zero() if mean is None else mean
        self.kernel = Kernel.squared_exponential() if kernel is None else kernel
        self.noise_variance = noise_variance
        self._X_train = self._y_train = None
        self._handle = None
        self._dev = None

    n_features = property(lambda s: len(s.features))

    def set_training_data(self, X, y):
        """gp.py: X (n_features x n), y (1 x n)."""
        X = np.atleast_2d(np.asarray(X.cpu() if isinstance(X, torch.Tensor) else X, dtype=float))
        y = np.atleast_2d(np.asarray(y.cpu() if isinstance(y, torch.Tensor) else y, dtype=float))
        if X.shape[0] != self.n_features:
            raise ValueError(f"Dimension mismatch. Supplied dimension for the features is {X.shape[0]}, but required "
                             f"dimension is {self.n_features}.")
        if y.shape[0] != 1:
            raise ValueError(f"Dimension mismatch. Supplied dimension for the labels is {y.shape[0]}, but required "
                             f"dimension is 1.")
        if X.shape[1] != y.shape[1]:
            raise ValueError("Number of observations in training matrix and target vector do not match!")
        self._X_train, self._y_train = X, y

    X_train = property(lambda s: s._X_train)

    def is_setup(self):
        """gp.py `is_setup`."""
        return getattr(self, '_handle', None) is not None

    y_train = property(lambda s: s._y_train)

    def setup(self, device_index=None, **kwargs):
        if self._X_train is None or self._y_train is None:
            raise RuntimeError("The training data has not been set. Please run the method set_training_data() to "
                               "proceed.")
        self._dev = device(device_index)
        nf, n = self._X_train.shape
        kp = np.asarray(self.kernel.program(nf), dtype=np.float64)
        mp = np.asarray(self.mean.program(nf), dtype=np.float64)
        X = np.ascontiguousarray(self._X_train)
        y = np.ascontiguousarray(self._y_train.ravel())
        h = C.c_void_p()
        _lib.check(_lib.lib().hilo_gp_create(self._dev.index, nf, n, X.ctypes.data, y.ctypes.data, kp.ctypes.data, kp.size,
                                             mp.ctypes.data, mp.size, float(self.noise_variance), C.byref(h)))
        self._destroy()
        self._handle = h

    def _destroy(self):
        if self._handle is not None:
            _lib.lib().hilo_gp_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def log_marginal_likelihood(self):
        """Log marginal likelihood of the training data plus the log densities of the hyper-priors (gp.py:553-559)."""
        if self._handle is None:
            raise RuntimeError("The GP has not been set up yet. Please run the setup() method before predicting.")
        v = C.c_double()
        _lib.check(_lib.lib().hilo_gp_log_marginal_likelihood(self._handle, C.byref(v)))
        return v.value + self._log_hyperprior()

    # ---- hyper-priors (util/probability.py:61-118, 171-217; gp.py:553-559) ----
    def set_hyperprior(self, name, prior, mean=0., variance=1., nu=None):
        """Prior on the hyper-parameter `name` (one of `hyperparameter_names`; `prior`: 'Gaussian', 'Laplace', 'Students_T' or
        None to remove it).  The reference attaches it as `parameter.prior = 'Laplace'; parameter.prior.mean = ...`; like
        there the density is evaluated at the optimisation variable - log(value), or log(value)/2 for a `*variance*`
        parameter - and its logarithm is added to the log marginal likelihood that `fit_model` maximises."""
        names = self.hyperparameter_names
        if name not in names:
            raise KeyError(f"'{name}' is not among the hyper-parameters {names}")
        if not hasattr(self, '_priors'):
            self._priors = {}
        if prior is None:
            self._priors.pop(name, None)
            return
        kind = str(prior).lower().replace("'", '').replace(' ', '_')
        if kind not in ('gaussian', 'laplace', 'students_t'):
            raise ValueError(f"Prior '{prior}' not recognized")
        if kind == 'students_t' and (nu is None or nu <= 2.):
            raise ValueError("The Student's t prior needs nu > 2")
        if not variance > 0.:
            raise ValueError("The variance of a prior must be positive")
        self._priors[name] = (kind, float(mean), float(variance), None if nu is None else float(nu))

    def _log_hyperprior(self):
        priors = getattr(self, '_priors', None)
        if not priors:
            return 0.
        from math import lgamma, log, pi, sqrt
        total = 0.
        for name, value in zip(self.hyperparameter_names, self.hyperparameter_values):
            if name not in priors:
                continue
            kind, mu, var, nu = priors[name]
            w = log(value) / 2. if 'variance' in name else log(value)
            if kind == 'gaussian':
                total += -(w - mu) ** 2 / (2. * var) - log(2. * pi * var) / 2.
            elif kind == 'laplace':
                b = sqrt(var / 2.)
                total += -abs(w - mu) / b - log(2. * b)
            else:
                total += lgamma((nu + 1.) / 2.) - lgamma(nu / 2.) - log(var * (nu - 2.) * pi) / 2. \
                    - (nu + 1.) / 2. * log(1. + (w - mu) ** 2 / (var * (nu - 2.)))
        return total

    def predict_quantiles(self, quantiles=None, X_query=None, mean=None, var=None):
        """gp.py:720-746: (lower, upper) = mean + Phi^-1(q / 100) sqrt(var + noise variance) for the two percentages in
        `quantiles` (default 2.5 / 97.5), from a noise-free prediction at `X_query` or from a given mean / variance."""
        from scipy.special import ndtri
        if quantiles is None:
            quantiles = (2.5, 97.5)
        if X_query is not None:
            mean, var = self.predict(X_query, noise_free=True)
        elif mean is None and var is None:
            return None
        tensor = isinstance(mean, torch.Tensor)
        m = mean.cpu().numpy() if tensor else np.asarray(mean, dtype=float)
        v = var.cpu().numpy() if isinstance(var, torch.Tensor) else np.asarray(var, dtype=float)
        out = [float(ndtri(q / 100.)) * np.sqrt(v + float(self.noise_variance)) + m for q in quantiles]
        if tensor:
            out = [torch.as_tensor(o, device=mean.device) for o in out]
        return out[0], out[1]

    # ---- hyper-parameters and their fit (gp.py:408-430, :660-697) ----
    def _handles(self):
        return [(self, 'noise_variance', None)] + self.kernel.hyperparameter_handles()

    @property
    def hyperparameter_names(self):
        return ['GP.noise_variance'] + self.kernel.hyperparameter_names

    @property
    def hyperparameter_values(self):
        """Values in the order of the reference's `gp.hyperparameters`: [noise variance | kernel hyper-parameters]."""
        return [float(getattr(o, a) if i is None else getattr(o, a)[i]) for o, a, i in self._handles()]

    def _set_hyperparameters(self, values):
        for (o, a, i), v in zip(self._handles(), values):
            if i is None:
                setattr(o, a, float(v))
            else:
                cur = list(getattr(o, a))
                cur[i] = float(v)
                setattr(o, a, cur)

    # ---- device services of fit_model (overridable: the CPU tests of the host driver serve them from the oracle) ----
    def _device_refit(self):
        """Factorise at the object's current hyper-parameters into the handle's buffers (hilo_gp_refit)."""
        kp = np.ascontiguousarray(self.kernel.program(self._X_train.shape[0]), dtype=np.float64)
        try:
            if self.mean.hyperparameter_handles():
                mp = np.ascontiguousarray(self.mean.program(self._X_train.shape[0]), dtype=np.float64)
                _lib.check(_lib.lib().hilo_gp_set_mean_program(self._handle, mp.ctypes.data, mp.size))
            _lib.check(_lib.lib().hilo_gp_refit(self._handle, kp.ctypes.data, kp.size, float(self.noise_variance)))
            return True
        except (_lib.NotPositiveDefinite, ValueError):
            return False

    def _device_lml_gradient(self, th, h):
        """d LML / d theta at the FREE log hyper-parameters `th` (the object is factorised there): hilo_gp_lml_gradient with
        the kernel programs / noise variances at th +- h e_i."""
        nf = self._X_train.shape[0]
        progs, noise = [], []
        for i in range(th.size):
            for sgn in (1., -1.):
                e = np.zeros_like(th)
                e[i] = sgn * h
                self._set_hyperparameters(np.exp(th + e))
                progs.append(np.ascontiguousarray(self.kernel.program(nf), dtype=np.float64))
                noise.append(float(self.noise_variance))
        self._set_hyperparameters(np.exp(th))
        progs = np.ascontiguousarray(np.stack(progs))
        noise = np.ascontiguousarray(np.array(noise))
        hh = np.full(th.size, float(h))
        out = np.zeros(th.size)
        _lib.check(_lib.lib().hilo_gp_lml_gradient(self._handle, th.size, progs.ctypes.data, noise.ctypes.data, hh.ctypes.data,
                                                   out.ctypes.data))
        return out

    def fit_model(self, gtol=1e-8, maxiter=500):
        """Optimises the hyper-parameters by minimising the negative log marginal likelihood (gp.py:660-697): noise variance and
        kernel hyper-parameters over their logarithms (kernel.py:127-130), the non-fixed hyper-parameters of the mean function
        over their values (`positive=False`, mean.py:278; fitted together with the kernel's, gp.py:408-414).  Every objective
        value is one device factorisation into the handle's buffers (`hilo_gp_refit`), the gradient with respect to the noise /
        kernel variables is the device trace formula 1/2 tr((alpha alpha^T - K^-1) dK/dtheta) (`hilo_gp_lml_gradient`, one
        factorisation for all of them), with respect to the few mean variables a central difference of the objective;
        quasi-Newton BFGS on the host - the reference hands the same objective to its NLP solver.  An indefinite trial point is
        +inf, never an exception; the object always ends on a factorised set of hyper-parameters.  Warns, like the reference,
        when the optimiser does not reach a stationary point."""
        if self._handle is None:
            raise RuntimeError("The GP has not been set up yet. Please run the setup() method before fitting.")
        import warnings
        from scipy.optimize import minimize
        dev_index = self._dev.index
        th_all = np.log(np.asarray(self.hyperparameter_values, dtype=float))
        if not np.all(np.isfinite(th_all)):
            raise ValueError("Hyper-parameters must be positive to be fitted in log space")
        # bounds of util/machine_learning.py:283-334: 'fixed' hyper-parameters stay out of the optimisation, numeric bounds box
        # the (log) variable; the noise variance is always free
        bnd = [None] + list(self.kernel.hyperparameter_bounds())
        free = [i for i, b in enumerate(bnd) if b != 'fixed']
        box = [(None if (bnd[i] is None or not bnd[i][0] > 0) else float(np.log(bnd[i][0])),
                None if (bnd[i] is None or not np.isfinite(bnd[i][1])) else float(np.log(bnd[i][1]))) for i in free]
        boxed = any(lo is not None or hi is not None for lo, hi in box)
        set_all = self._set_hyperparameters
        # hyper-parameters of the mean function: optimisation variable = the value
        mh = self.mean.hyperparameter_handles() if isinstance(self.mean, Mean) else []
        n_log, n_mean = len(free), len(mh)

        def set_mean(values):
            for (o, a, i), v in zip(mh, values):
                if i is None:
                    setattr(o, a, float(v))
                else:
                    cur = list(getattr(o, a))
                    cur[i] = float(v)
                    setattr(o, a, cur)
        mv0 = np.array([float(getattr(o, a) if i is None else getattr(o, a)[i]) for o, a, i in mh])
        box = box + [(None, None)] * n_mean

        def set_free(values):
            full = np.exp(th_all)
            full[free] = values
            set_all(full)
        self._set_hyperparameters = set_free            # the closures below only see the free ones
        x0 = np.concatenate([th_all[free], mv0])

        h_step = 1e-5
        state = {'x': None, 'ok': False}

        def refit(x):
            """One device factorisation into the handle's buffers; False at an indefinite trial point."""
            self._set_hyperparameters(np.exp(x[:n_log]))
            set_mean(x[n_log:])
            state['x'], state['ok'] = np.array(x), bool(self._device_refit())
            return state['ok']

        def f(x):
            if not refit(x):
                return np.inf
            v = -self.log_marginal_likelihood()
            return v if np.isfinite(v) else np.inf

        def g(x):
            """-(d LML / d x): device trace formula for the noise / kernel variables + the hyper-priors' part by central
            differences of their closed-form log densities; central differences of the objective for the mean variables."""
            if state['x'] is None or not np.array_equal(state['x'], x):
                refit(x)
            if not state['ok']:
                return np.zeros_like(x)
            th = x[:n_log]
            out = np.zeros_like(x)
            out[:n_log] = self._device_lml_gradient(th, h_step)
            if getattr(self, '_priors', None):
                for i in range(n_log):
                    e = np.zeros_like(th)
                    e[i] = h_step
                    self._set_hyperparameters(np.exp(th + e))
                    lp = self._log_hyperprior()
                    self._set_hyperparameters(np.exp(th - e))
                    out[i] += (lp - self._log_hyperprior()) / (2 * h_step)
            self._set_hyperparameters(np.exp(th))
            for i in range(n_mean):
                hm = 1e-6 * max(1., abs(x[n_log + i]))
                e = np.zeros_like(x)
                e[n_log + i] = hm
                up, dn = f(x + e), f(x - e)
                out[n_log + i] = -(up - dn) / (2 * hm) if np.isfinite(up) and np.isfinite(dn) else 0.
            if n_mean:
                refit(x)
            return -out
        good = x0.copy()
        try:
            if boxed:
                res = minimize(f, x0, jac=g, method='L-BFGS-B', bounds=box, options={'gtol': gtol, 'ftol': 1e-15, 'maxiter': maxiter})
            else:
                res = minimize(f, x0, jac=g, method='BFGS', options={'gtol': gtol, 'maxiter': maxiter})
            good = res.x
        finally:
            # whatever happened inside the optimiser: the object ends on a consistent, factorised set of hyper-parameters
            self._set_hyperparameters(np.exp(good[:n_log]))
            set_mean(good[n_log:])
            self.setup(device_index=dev_index)
            del self._set_hyperparameters                # back to the class's setter of all hyper-parameters
        self._set_hyperparameters = set_free
        gr = g(res.x)
        del self._set_hyperparameters
        if boxed:                                        # projected gradient: a bound that is active does not count
            for i, (lo, hi) in enumerate(box):
                if (lo is not None and res.x[i] <= lo + 1e-12 and gr[i] > 0) or (hi is not None and res.x[i] >= hi - 1e-12 and gr[i] < 0):
                    gr[i] = 0.
        gn = float(np.max(np.abs(gr))) if gr.size else 0.
        self._optimization_stats = {'success': bool(res.success or gn < 1e-4), 'message': str(res.message),
                                    'iter_count': int(res.nit), 'max_gradient': gn, 'fun': float(res.fun)}
        set_free(np.exp(res.x[:n_log]))
        set_mean(res.x[n_log:])
        self.setup(device_index=dev_index)
        if not self._optimization_stats['success']:
            warnings.warn(f"Fitting of GP didn't terminate successfully\nSolver message: {res.message}\n"
                          f"Try to use a different solver")

    def predict(self, X_query, noise_free=False, return_var=True):
        """gp.py:699-718: returns (mean (1 x m), var (1 x m)); numpy in -> numpy out, device tensor in -> tensor out."""
        if self._handle is None:
            raise RuntimeError("The GP has not been set up yet. Please run the setup() method before predicting.")
        host = not isinstance(X_query, torch.Tensor)
        Xq = to_dev(np.atleast_2d(X_query) if host else X_query, self._dev)
        if Xq.ndim == 1:
            Xq = Xq.reshape(-1, 1)
        if Xq.shape[0] != self.n_features:
            raise ValueError(f"Dimension mismatch. Supplied dimension for the features is {Xq.shape[0]}, but required "
                             f"dimension is {self.n_features}.")
        m = Xq.shape[1]
        mean = torch.empty(1, m, dtype=torch.float64, device=self._dev)
        var = torch.empty(1, m, dtype=torch.float64, device=self._dev) if return_var else None
        _lib.check(_lib.lib().hilo_gp_predict(self._handle, m, ptr(Xq), int(bool(noise_free)), ptr(mean), ptr(var),
                                              stream_ptr(self._dev)))
        if host:
            return mean.cpu().numpy(), (var.cpu().numpy() if return_var else None)
        return mean, var
