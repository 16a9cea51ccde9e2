"""Small expression builder: the stand-in for the CasADi SX expressions the reference accepts for path references
(`nmpc.create_path_variable()` -> `ref=ca.vertcat(sin(theta), ...)`, hilo_mpc/util/modeling.py:252-261) and for
nonlinear constraints (`nmpc.stage_constraint.constraint = ...`, modeling.py:930-940).

    vx, vy = model.x['vx'], model.x['vy']          # symbols of the zoo model
    nmpc.stage_constraint.constraint = vx ** 2 + vy ** 2
    theta = nmpc.create_path_variable()
    nmpc.quad_stage_cost.add_states(names=['px', 'py'], weights=[10, 10], ref=[sin(theta), sin(2 * theta)],
                                    path_following=True)

An expression compiles to the flat postfix program of include/hilo_hip.h (HILO_X_*), interpreted on the device for
values and second-order Taylor numbers alike (hilo_mpc_amd/csrc/hilo_expr.h).
"""
import numbers

X_CONST, X_VARX, X_VARU, X_VARP = 0, 1, 2, 3
X_ADD, X_SUB, X_MUL, X_DIV, X_NEG, X_SQ, X_SIN, X_COS, X_EXP, X_LOG, X_SQRT, X_POWI = 10, 11, 12, 13, 14, 15, 16, 17, 18, \
    19, 20, 21
STACK = 8


class Expr:
    """Node of an expression tree: op in {'const','x','u','p','z','theta','t', binary / unary op names, 'gp','gpd','gpvar','gpk'}.
    Unary functions: sq sin cos exp log sqrt (device interpreter and compiled code) and - compiled code only - the rest of the
    reference's table (util/parsing.py:36-58): log10 fabs sign asin acos atan asinh acosh atanh; binary: atan2."""
    __slots__ = ('op', 'args', 'value', 'name', 'serial')
    _count = 0

    def __init__(self, op, args=(), value=None, name=None):
        self.op, self.args, self.value, self.name = op, tuple(args), value, name
        Expr._count += 1
        self.serial = Expr._count        # creation order = the order the user's statements were evaluated (code emission)

    # ---- operators -----------------------------------------------------------------------------------------
    @staticmethod
    def wrap(v):
        if isinstance(v, Expr):
            return v
        if isinstance(v, numbers.Real):
            return Expr('const', value=float(v))
        raise TypeError(f"cannot use {type(v).__name__} in an expression")

    def __add__(self, o): return Expr('add', (self, Expr.wrap(o)))
    def __radd__(self, o): return Expr('add', (Expr.wrap(o), self))
    def __sub__(self, o): return Expr('sub', (self, Expr.wrap(o)))
    def __rsub__(self, o): return Expr('sub', (Expr.wrap(o), self))
    def __mul__(self, o): return Expr('mul', (self, Expr.wrap(o)))
    def __rmul__(self, o): return Expr('mul', (Expr.wrap(o), self))
    def __truediv__(self, o): return Expr('div', (self, Expr.wrap(o)))
    def __rtruediv__(self, o): return Expr('div', (Expr.wrap(o), self))
    def __neg__(self): return Expr('neg', (self,))
    def __pos__(self): return self

    def __pow__(self, n):
        if isinstance(n, numbers.Real) and float(n) == int(n) and abs(int(n)) <= 16:
            n = int(n)
            return Expr('sq', (self,)) if n == 2 else Expr('powi', (self,), value=n)
        if isinstance(n, numbers.Real) and float(n) == 0.5:
            return Expr('sqrt', (self,))
        if isinstance(n, (numbers.Real, Expr)):
            # general power of a POSITIVE base through the operations the device has: a^b = exp(b log a)
            return Expr('exp', (Expr.wrap(n) * Expr('log', (self,)),))
        raise NotImplementedError("the exponent must be a number or an expression")

    def __rpow__(self, a):
        return Expr('exp', (self * Expr('log', (Expr.wrap(a),)),))

    def __repr__(self):
        if self.op == 'const':
            return repr(self.value)
        if self.op in ('x', 'u', 'p', 'z', 'theta', 't', 'dt'):
            return self.name
        return f"{self.op}({', '.join(map(repr, self.args))})"

    # ---- substitution --------------------------------------------------------------------------------------
    def nodes(self, out=None):
        """All distinct nodes below (and including) this one, keyed by id."""
        out = {} if out is None else out
        stack = [self]
        while stack:
            n = stack.pop()
            if id(n) not in out:
                out[id(n)] = n
                stack.extend(n.args)
        return out

    @staticmethod
    def substitute(exprs, leaf):
        """Rebuilds `exprs` with every leaf replaced by `leaf(node)` (None = keep).  Sharing is preserved and the new nodes
        are created in the creation order of the old ones, so the emitted code keeps the order of the user's statements."""
        allnodes = {}
        for e in exprs:
            e.nodes(allnodes)
        new = {}
        for n in sorted(allnodes.values(), key=lambda q: q.serial):
            if n.args:
                a = tuple(new[id(c)] for c in n.args)
                new[id(n)] = n if all(x is y for x, y in zip(a, n.args)) else Expr(n.op, a, n.value, n.name)
            else:
                r = leaf(n)
                new[id(n)] = n if r is None else r
        return [new[id(e)] for e in exprs]

    # ---- compilation ---------------------------------------------------------------------------------------
    def depends_on(self, kind):
        return self.op == kind or any(a.depends_on(kind) for a in self.args)

    def _emit(self, out, theta_index):
        for a in self.args:
            a._emit(out, theta_index)
        op = self.op
        if op == 'const':
            out += [X_CONST, self.value]
        elif op == 'x':
            out += [X_VARX, float(self.value)]
        elif op == 'u':
            out += [X_VARU, float(self.value)]
        elif op == 'p':
            out += [X_VARP, float(self.value)]
        elif op == 'theta':
            if theta_index is None:
                raise ValueError("a path variable can only appear in path references")
            out += [X_VARX, float(theta_index)]
        elif op == 'powi':
            out += [X_POWI, float(self.value)]
        elif op in ('gp', 'gpvar', 'gpd', 'gpk'):
            raise ValueError("a learned term cannot be evaluated by the device interpreter (run-time compiled models only)")
        elif op == 'z':
            raise ValueError("an algebraic state cannot be evaluated by the device interpreter (run-time compiled models only)")
        elif op in NATIVE_UNARY or op == 'atan2':
            raise ValueError(f"'{op}' is a function of the compiled code only (run-time compiled models), not of the device interpreter")
        else:
            out += [{'add': X_ADD, 'sub': X_SUB, 'mul': X_MUL, 'div': X_DIV, 'neg': X_NEG, 'sq': X_SQ, 'sin': X_SIN,
                     'cos': X_COS, 'exp': X_EXP, 'log': X_LOG, 'sqrt': X_SQRT}[op], 0.]

    def depth(self):
        """Stack slots the postfix evaluation needs."""
        if not self.args:
            return 1
        d = [a.depth() for a in self.args]
        return max(d[0], 1 + d[1]) if len(d) == 2 else d[0]

    def program(self, theta_index=None):
        """[len, (op, arg)...] as a list of floats."""
        if self.depth() > STACK:
            raise ValueError(f"expression too deep for the device interpreter (stack of {STACK}); re-associate it")
        code = []
        self._emit(code, theta_index)
        return [float(len(code))] + [float(c) for c in code]


def _unary(op):
    def f(a):
        return Expr(op, (Expr.wrap(a),))
    f.__name__ = op
    return f


sin, cos, exp, log, sqrt = (_unary(n) for n in ('sin', 'cos', 'exp', 'log', 'sqrt'))
# the rest of the reference's function table (util/parsing.py:36-58 -> ca.log10, ca.sign, ca.fabs, ca.asin, ...): native device
# functions of the compiled code (csrc/hilo_ad.h), not of the postfix interpreter
NATIVE_UNARY = ('log10', 'fabs', 'sign', 'asin', 'acos', 'atan', 'asinh', 'acosh', 'atanh')
log10, fabs, sign, asin, acos, atan, asinh, acosh, atanh = (_unary(n) for n in NATIVE_UNARY)
arcsin, arccos, arctan, arsinh, arcosh, artanh = asin, acos, atan, asinh, acosh, atanh


def atan2(y, x):
    return Expr('atan2', (Expr.wrap(y), Expr.wrap(x)))


arctan2 = atan2


def fmin(a, b):
    """min(a, b) = (a + b - |a - b|) / 2: value and derivatives (CasADi's: the active branch, the mean at a tie) from the parts."""
    a, b = Expr.wrap(a), Expr.wrap(b)
    return 0.5 * (a + b - fabs(a - b))


def fmax(a, b):
    a, b = Expr.wrap(a), Expr.wrap(b)
    return 0.5 * (a + b + fabs(a - b))


# functions composed of the device's operations (no new device code; derivatives follow from the parts)
def tan(a):
    a = Expr.wrap(a)
    return sin(a) / cos(a)


def sinh(a):
    e = exp(Expr.wrap(a))
    return 0.5 * (e - 1.0 / e)


def cosh(a):
    e = exp(Expr.wrap(a))
    return 0.5 * (e + 1.0 / e)


def tanh(a):
    e2 = exp(2.0 * Expr.wrap(a))
    return (e2 - 1.0) / (e2 + 1.0)


class SymVector:
    """`model.x` / `model.u` / `model.p`: symbols addressable by position or by name."""

    def __init__(self, kind, names):
        self._kind, self._names = kind, list(names)
        self._syms = [Expr(kind, value=i, name=n) for i, n in enumerate(self._names)]

    def __getitem__(self, key):
        if isinstance(key, str):
            if key not in self._names:
                raise KeyError(f"'{key}' is not among {self._names}")
            return self._syms[self._names.index(key)]
        return self._syms[key]

    def __len__(self):
        return len(self._syms)

    def __iter__(self):
        return iter(self._syms)


def _is_c(e, v=None):
    return e.op == 'const' and (v is None or e.value == v)


def _add(a, b):
    if _is_c(a, 0.0):
        return b
    if _is_c(b, 0.0):
        return a
    if _is_c(a) and _is_c(b):
        return Expr.wrap(a.value + b.value)
    return a + b


def _mul(a, b):
    if _is_c(a, 0.0) or _is_c(b, 0.0):
        return Expr.wrap(0.0)
    if _is_c(a, 1.0):
        return b
    if _is_c(b, 1.0):
        return a
    if _is_c(a) and _is_c(b):
        return Expr.wrap(a.value * b.value)
    return a * b


# derivative of the native unary functions as expressions of their argument
_D_UNARY = {
    'log10': lambda a: 0.4342944819032518 / a,
    'fabs': lambda a: sign(a),
    'sign': lambda a: Expr.wrap(0.0),
    'asin': lambda a: 1.0 / sqrt(1.0 - a * a),
    'acos': lambda a: -1.0 / sqrt(1.0 - a * a),
    'atan': lambda a: 1.0 / (1.0 + a * a),
    'asinh': lambda a: 1.0 / sqrt(a * a + 1.0),
    'acosh': lambda a: 1.0 / sqrt(a * a - 1.0),
    'atanh': lambda a: 1.0 / (1.0 - a * a),
}


def diff(e, var, memo=None):
    """d e / d var as an expression tree (`ca.jacobian` of the reference where a derivative becomes part of a MODEL, e.g. the
    covariance propagation of the stochastic NMPC, mpc.py:2534-2575); `var` is a leaf (a state, input or parameter symbol,
    matched by kind and index).  Zero and unit factors are removed; shared sub-expressions share their derivative.  The
    derivatives the SOLVER needs are not built here - the compiled code is evaluated in Taylor / dual arithmetic."""
    memo = {} if memo is None else memo
    e = Expr.wrap(e)
    order = sorted(e.nodes().values(), key=lambda q: q.serial)
    zero, one = Expr.wrap(0.0), Expr.wrap(1.0)
    for n in order:
        if id(n) in memo:
            continue
        op, a = n.op, n.args
        d = [memo[id(c)] for c in a]
        if op in ('x', 'u', 'p', 'z', 'theta'):
            r = one if (op == var.op and n.value == var.value) else zero
        elif op in ('const', 'dt', 't'):
            r = zero
        elif op == 'add':
            r = _add(d[0], d[1])
        elif op == 'sub':
            r = d[0] if _is_c(d[1], 0.0) else (Expr('neg', (d[1],)) if _is_c(d[0], 0.0) else d[0] - d[1])
        elif op == 'neg':
            r = zero if _is_c(d[0], 0.0) else -d[0]
        elif op == 'mul':
            r = _add(_mul(d[0], a[1]), _mul(a[0], d[1]))
        elif op == 'div':                                  # (a / b)' = a' / b - (a / b) b' / b
            r = zero if _is_c(d[0], 0.0) else d[0] / a[1]
            if not _is_c(d[1], 0.0):
                t = _mul(n, d[1]) / a[1]
                r = -t if _is_c(r, 0.0) else r - t
        elif op == 'sq':
            r = _mul(_mul(Expr.wrap(2.0), a[0]), d[0])
        elif op == 'powi':
            k = int(n.value)
            r = zero if k == 0 else _mul(_mul(Expr.wrap(float(k)), a[0] ** (k - 1) if k != 1 else one), d[0])
        elif op == 'sin':
            r = _mul(cos(a[0]), d[0])
        elif op == 'cos':
            r = _mul(-sin(a[0]), d[0])
        elif op == 'exp':
            r = _mul(n, d[0])
        elif op == 'log':
            r = zero if _is_c(d[0], 0.0) else d[0] / a[0]
        elif op == 'sqrt':
            r = zero if _is_c(d[0], 0.0) else d[0] / (2.0 * n)
        elif op in _D_UNARY:
            r = zero if _is_c(d[0], 0.0) else _mul(_D_UNARY[op](a[0]), d[0])
        elif op == 'atan2':                                # d atan2(y, x) = (x dy - y dx) / (x^2 + y^2)
            num = _add(_mul(a[1], d[0]), -_mul(a[0], d[1]) if not _is_c(d[1], 0.0) else zero)
            r = zero if _is_c(num, 0.0) else num / (a[0] * a[0] + a[1] * a[1])
        elif op == 'gp':                                   # chain rule through the posterior mean: sum_j dmean/dfeature_j * feature_j'
            r = zero
            for j, dj in enumerate(d):
                if not _is_c(dj, 0.0):
                    r = _add(r, _mul(Expr('gpd', a, value=(int(n.value), j), name=n.name), dj))
        else:
            raise NotImplementedError(f"no expression-level derivative for operator '{op}'")
        memo[id(n)] = r
    return memo[id(e)]


def jacobian(exprs, variables):
    """[[d e_i / d v_j]] for lists of expressions and leaf symbols."""
    exprs = [Expr.wrap(e) for e in exprs]
    cols = []
    for v in variables:
        memo = {}                                          # one memo per variable: rows share their common sub-expressions
        cols.append([diff(e, v, memo) for e in exprs])
    return [[cols[j][i] for j in range(len(variables))] for i in range(len(exprs))]


def compile_block(exprs, theta_index=None):
    """Programs of several expressions back to back (the layout `hilo_nmpc_desc.path_prog` / `con_prog` expect)."""
    out = []
    for e in exprs:
        out += Expr.wrap(e).program(theta_index)
    return out


def hessian_structure(e):
    """(dep, pairs) of an expression: the leaves it depends on and the leaf pairs (a, b), a <= b in a fixed order of the keys
    (kind, index), whose second derivative can be non-zero - the usual conservative propagation (a product couples the
    dependencies of its factors, a nonlinear function those of its argument with themselves).  Used to drop Taylor directions of
    the interior-point engine's derivative phase whose Hessian entry is structurally zero (csrc/hilo_ocp.h `pair_mask`)."""
    e = Expr.wrap(e)
    dep, prs = {}, {}

    def cross(A, B):
        return {(a, b) if a <= b else (b, a) for a in A for b in B}

    for n in sorted(e.nodes().values(), key=lambda q: q.serial):
        op, a = n.op, n.args
        if op in ('x', 'u', 'p', 'z', 'theta'):
            d, p = {(op, int(n.value) if n.value is not None else 0)}, set()
        elif not a:
            d, p = set(), set()
        else:
            ds, ps = [dep[id(c)] for c in a], [prs[id(c)] for c in a]
            d = set().union(*ds)
            p = set().union(*ps)
            if op in ('add', 'sub', 'neg'):
                pass
            elif op == 'mul':
                p |= cross(ds[0], ds[1])
            elif op == 'div':
                p |= cross(ds[0], ds[1]) | cross(ds[1], ds[1])
            elif op == 'powi' and int(n.value) in (0, 1):
                pass
            else:                                         # sq, powi, every unary function, atan2, learned terms: all pairs
                p |= cross(d, d)
        dep[id(n)], prs[id(n)] = d, p
    return dep[id(e)], prs[id(e)]
