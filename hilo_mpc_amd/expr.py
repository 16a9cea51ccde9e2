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
    """Node of an expression tree: op in {'const','x','u','p','theta', binary / unary op names}."""
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
        raise NotImplementedError("only integer powers in [-16, 16] and 0.5 are available on the device")

    def __repr__(self):
        if self.op == 'const':
            return repr(self.value)
        if self.op in ('x', 'u', 'p', 'z', 'theta'):
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
        elif op == 'gp':
            raise ValueError("a learned term cannot be evaluated by the device interpreter (run-time compiled models only)")
        elif op == 'z':
            raise ValueError("an algebraic state cannot be evaluated by the device interpreter (run-time compiled models only)")
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


def compile_block(exprs, theta_index=None):
    """Programs of several expressions back to back (the layout `hilo_nmpc_desc.path_prog` / `con_prog` expect)."""
    out = []
    for e in exprs:
        out += Expr.wrap(e).program(theta_index)
    return out
