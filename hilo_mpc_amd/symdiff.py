"""Symbolic first and second derivatives of model right-hand sides, emitted as straight-line HIP code.

The reference gets the Hessian of the Lagrangian from CasADi's symbolic differentiation of the SX graph of the discretised
model (`ca.nlpsol` with the exact Hessian, hilo_mpc/modules/controller/mpc.py:1778-1787): common sub-expressions are shared and
structural zeros never computed.  The counterpart here for the interior-point engine's derivative phase
(csrc/hilo_ocp.h::eval_derivs_sym): for a right-hand side f(x, u, p) written as expressions (hilo_mpc_amd/expr.py) this module
produces

    jx(x, u, p, fx)              df/dx                                             (adjoint sweep through the Runge-Kutta stages)
    jh(x, u, p, kb, J, H)        J = df/d(x,u), H = sum_m kb[m] d2 f_m / d(x,u)^2     (packed lower triangle)

on a hash-consed DAG (one node per distinct sub-expression, constants folded, x*0 / x*1 / x+0 removed), so the emitted code
evaluates every shared term once.  The engine pushes J through the stages with the chain rule and obtains the Hessian of
lambda^T Phi from the second-order adjoint  sum_i dW_i^T H_i(kbar_i) dW_i  - no second-order Taylor sweeps per direction pair.
"""
import math

from .expr import Expr


# numeric counterparts of the unary functions (constant folding, evaluate())
_NUMERIC = {'sin': math.sin, 'cos': math.cos, 'exp': math.exp, 'log': math.log, 'sqrt': math.sqrt, 'log10': math.log10,
            'fabs': math.fabs, 'sign': lambda v: (v > 0) - (v < 0) + 0.0, 'asin': math.asin, 'acos': math.acos, 'atan': math.atan,
            'asinh': math.asinh, 'acosh': math.acosh, 'atanh': math.atanh}


class Dag:
    """Hash-consed expression DAG with local simplification.  Nodes are integers; `self.nodes[i] = (op, a, b, value)`."""

    def __init__(self):
        self.nodes, self.index, self._d = [], {}, {}

    def _mk(self, op, a=-1, b=-1, value=None):
        key = (op, a, b, value)
        i = self.index.get(key)
        if i is None:
            i = len(self.nodes)
            self.nodes.append(key)
            self.index[key] = i
        return i

    # ---- constructors with simplification ---------------------------------------------------------------
    def const(self, v):
        v = float(v)
        return self._mk('const', value=0.0 if v == 0.0 else v)       # -0.0 -> 0.0

    def var(self, kind, idx):
        return self._mk(kind, value=int(idx))

    def is_const(self, i, v=None):
        n = self.nodes[i]
        return n[0] == 'const' and (v is None or n[3] == v)

    def cval(self, i):
        return self.nodes[i][3]

    def add(self, a, b):
        if self.is_const(a) and self.is_const(b):
            return self.const(self.cval(a) + self.cval(b))
        if self.is_const(a, 0.0):
            return b
        if self.is_const(b, 0.0):
            return a
        if self.nodes[b][0] == 'neg':
            return self.sub(a, self.nodes[b][1])
        if self.nodes[a][0] == 'neg':
            return self.sub(b, self.nodes[a][1])
        if a > b and not self.is_const(a) and not self.is_const(b):
            a, b = b, a                                                # commutative: one canonical order
        return self._mk('add', a, b)

    def sub(self, a, b):
        if self.is_const(a) and self.is_const(b):
            return self.const(self.cval(a) - self.cval(b))
        if self.is_const(b, 0.0):
            return a
        if self.is_const(a, 0.0):
            return self.neg(b)
        if a == b:
            return self.const(0.0)
        if self.nodes[b][0] == 'neg':
            return self.add(a, self.nodes[b][1])
        return self._mk('sub', a, b)

    def neg(self, a):
        if self.is_const(a):
            return self.const(-self.cval(a))
        if self.nodes[a][0] == 'neg':
            return self.nodes[a][1]
        return self._mk('neg', a)

    def mul(self, a, b):
        if self.is_const(a) and self.is_const(b):
            return self.const(self.cval(a) * self.cval(b))
        for p, q in ((a, b), (b, a)):
            if self.is_const(p, 0.0):
                return self.const(0.0)
            if self.is_const(p, 1.0):
                return q
            if self.is_const(p, -1.0):
                return self.neg(q)
        if self.nodes[a][0] == 'neg' and self.nodes[b][0] == 'neg':
            return self.mul(self.nodes[a][1], self.nodes[b][1])
        if self.nodes[a][0] == 'neg':
            return self.neg(self.mul(self.nodes[a][1], b))
        if self.nodes[b][0] == 'neg':
            return self.neg(self.mul(a, self.nodes[b][1]))
        # constant * (constant * x) -> one constant
        for p, q in ((a, b), (b, a)):
            if self.is_const(p) and self.nodes[q][0] == 'mul' and self.is_const(self.nodes[q][1]):
                return self.mul(self.const(self.cval(p) * self.cval(self.nodes[q][1])), self.nodes[q][2])
        if self.is_const(b) or (a > b and not self.is_const(a)):
            a, b = b, a                                                # constants first, otherwise canonical order
        return self._mk('mul', a, b)

    def div(self, a, b):
        if self.is_const(b):
            return self.mul(self.const(1.0 / self.cval(b)), a)
        if self.is_const(a, 0.0):
            return self.const(0.0)
        return self.mul(a, self.recip(b))

    def recip(self, a):
        if self.is_const(a):
            return self.const(1.0 / self.cval(a))
        if self.nodes[a][0] == 'neg':
            return self.neg(self.recip(self.nodes[a][1]))
        return self._mk('recip', a)

    def fun(self, op, a):
        if self.is_const(a):
            return self.const(_NUMERIC[op](self.cval(a)))
        return self._mk(op, a)

    def atan2(self, y, x):
        if self.is_const(y) and self.is_const(x):
            return self.const(math.atan2(self.cval(y), self.cval(x)))
        return self._mk('atan2', y, x)

    def sq(self, a):
        return self.mul(a, a)

    def powi(self, a, n):
        if n == 0:
            return self.const(1.0)
        if n < 0:
            return self.recip(self.powi(a, -n))
        r = a
        for _ in range(n - 1):
            r = self.mul(r, a)
        return r

    # ---- import of an expression tree -----------------------------------------------------------------------
    def from_expr(self, e, memo=None):
        memo = {} if memo is None else memo
        e = Expr.wrap(e)
        todo = sorted(e.nodes().values(), key=lambda q: q.serial)
        for n in todo:
            if id(n) in memo:
                continue
            a = [memo[id(c)] for c in n.args]
            op = n.op
            if op == 'const':
                r = self.const(n.value)
            elif op in ('x', 'u', 'p', 'z'):
                r = self.var(op, n.value)
            elif op == 'add':
                r = self.add(*a)
            elif op == 'sub':
                r = self.sub(*a)
            elif op == 'mul':
                r = self.mul(*a)
            elif op == 'div':
                r = self.div(*a)
            elif op == 'neg':
                r = self.neg(a[0])
            elif op == 'sq':
                r = self.sq(a[0])
            elif op == 'powi':
                r = self.powi(a[0], int(n.value))
            elif op in _NUMERIC:
                r = self.fun(op, a[0])
            elif op == 'atan2':
                r = self.atan2(a[0], a[1])
            else:
                raise NotImplementedError(f"no symbolic derivative for operator '{op}'")
            memo[id(n)] = r
        return memo[id(e)]

    # ---- differentiation ------------------------------------------------------------------------------------------
    def diff(self, i, wrt):
        """d node_i / d wrt, `wrt` a variable node."""
        key = (i, wrt)
        r = self._d.get(key)
        if r is not None:
            return r
        op, a, b, v = self.nodes[i]
        if i == wrt:
            r = self.const(1.0)
        elif op in ('const', 'x', 'u', 'p', 'z', 'kb'):
            r = self.const(0.0)
        elif op == 'add':
            r = self.add(self.diff(a, wrt), self.diff(b, wrt))
        elif op == 'sub':
            r = self.sub(self.diff(a, wrt), self.diff(b, wrt))
        elif op == 'neg':
            r = self.neg(self.diff(a, wrt))
        elif op == 'mul':
            r = self.add(self.mul(self.diff(a, wrt), b), self.mul(a, self.diff(b, wrt)))
        elif op == 'recip':                                            # d(1/a) = -(1/a)^2 da
            r = self.neg(self.mul(self.mul(i, i), self.diff(a, wrt)))
        elif op == 'sin':
            r = self.mul(self.fun('cos', a), self.diff(a, wrt))
        elif op == 'cos':
            r = self.neg(self.mul(self.fun('sin', a), self.diff(a, wrt)))
        elif op == 'exp':
            r = self.mul(i, self.diff(a, wrt))
        elif op == 'log':
            r = self.mul(self.recip(a), self.diff(a, wrt))
        elif op == 'sqrt':                                             # d sqrt(a) = da / (2 sqrt(a))
            r = self.mul(self.mul(self.const(0.5), self.recip(i)), self.diff(a, wrt))
        elif op == 'log10':
            r = self.mul(self.mul(self.const(0.4342944819032518), self.recip(a)), self.diff(a, wrt))
        elif op == 'fabs':                                             # CasADi: d|a| = sign(a) da
            r = self.mul(self.fun('sign', a), self.diff(a, wrt))
        elif op == 'sign':
            r = self.const(0.0)
        elif op in ('asin', 'acos'):                                   # +- da / sqrt(1 - a^2)
            g = self.recip(self.fun('sqrt', self.sub(self.const(1.0), self.mul(a, a))))
            r = self.mul(g if op == 'asin' else self.neg(g), self.diff(a, wrt))
        elif op == 'atan':
            r = self.mul(self.recip(self.add(self.const(1.0), self.mul(a, a))), self.diff(a, wrt))
        elif op in ('asinh', 'acosh'):                                 # da / sqrt(a^2 +- 1)
            inner = self.add(self.mul(a, a), self.const(1.0)) if op == 'asinh' else self.sub(self.mul(a, a), self.const(1.0))
            r = self.mul(self.recip(self.fun('sqrt', inner)), self.diff(a, wrt))
        elif op == 'atanh':
            r = self.mul(self.recip(self.sub(self.const(1.0), self.mul(a, a))), self.diff(a, wrt))
        elif op == 'atan2':                                            # (x dy - y dx) / (x^2 + y^2), node = atan2(a = y, b = x)
            num = self.sub(self.mul(b, self.diff(a, wrt)), self.mul(a, self.diff(b, wrt)))
            r = self.mul(num, self.recip(self.add(self.mul(a, a), self.mul(b, b))))
        else:
            raise NotImplementedError(op)
        self._d[key] = r
        return r

    # ---- emission --------------------------------------------------------------------------------------------------------
    def emit(self, outputs, names=('x', 'u', 'p'), generic=False):
        """Straight-line statements for the nodes `outputs` depend on; returns (lines, {node: C expression}).  generic: for a
        templated scalar type (`const auto`, reciprocal as 1.0 / x) instead of double."""
        nm = dict(zip(('x', 'u', 'p'), names), kb='kb', z='z')
        ctype = 'auto' if generic else 'double'
        need, stack = set(), [o for o in outputs]
        while stack:
            i = stack.pop()
            if i in need:
                continue
            need.add(i)
            op, a, b, _ = self.nodes[i]
            if a >= 0:
                stack.append(a)
            if b >= 0:
                stack.append(b)
        ref, lines = {}, []
        for i in sorted(need):                                         # children always have smaller numbers than parents
            op, a, b, v = self.nodes[i]
            if op == 'const':
                s = repr(float(v))
                ref[i] = s if ('e' in s or '.' in s or 'n' in s) else s + '.0'
                if v < 0:
                    ref[i] = f"({ref[i]})"
                continue
            if op in ('x', 'u', 'p', 'kb', 'z'):
                ref[i] = f"{nm[op]}[{v}]"
                continue
            if op in ('add', 'sub', 'mul'):
                rhs = f"{ref[a]} {'+' if op == 'add' else '-' if op == 'sub' else '*'} {ref[b]}"
            elif op == 'neg':
                rhs = f"-{ref[a]}"
            elif op == 'recip':
                rhs = f"1.0 / {ref[a]}" if generic else f"rcp_fast({ref[a]})"   # csrc/hilo_ad.h: v_rcp_f64 + two Newton steps
            elif op == 'atan2':
                rhs = f"atan2({ref[a]}, {ref[b]})"
            else:
                rhs = f"{op}({ref[a]})"
            ref[i] = f"s{len(lines)}"
            lines.append(f"    const {ctype} {ref[i]} = {rhs};")
        return lines, ref


    # ---- numeric evaluation (tests: the derivative DAG against the oracle's sympy derivatives) ---------------------------------
    def evaluate(self, outputs, x, u, p, kb=(), z=()):
        val = {}
        env = {'x': x, 'u': u, 'p': p, 'kb': kb, 'z': z}
        fn = _NUMERIC
        need, stack = set(), list(outputs)
        while stack:
            i = stack.pop()
            if i not in need:
                need.add(i)
                stack += [c for c in self.nodes[i][1:3] if c >= 0]
        for i in sorted(need):
            op, a, b, v = self.nodes[i]
            if op == 'const':
                val[i] = v
            elif op in env:
                val[i] = float(env[op][v])
            elif op == 'add':
                val[i] = val[a] + val[b]
            elif op == 'sub':
                val[i] = val[a] - val[b]
            elif op == 'mul':
                val[i] = val[a] * val[b]
            elif op == 'neg':
                val[i] = -val[a]
            elif op == 'recip':
                val[i] = 1.0 / val[a]
            elif op == 'atan2':
                val[i] = math.atan2(val[a], val[b])
            else:
                val[i] = fn[op](val[a])
        return [val[o] for o in outputs]


def derivative_dag(n_x, n_u, ode):
    """(dag, f, J, H, kb): J[m][j] = d f_m / d w_j, H packed lower triangle of sum_m kb[m] d2 f_m / dw2, w = (x, u)."""
    g = Dag()
    memo = {}
    f = [g.from_expr(e, memo) for e in ode]
    nz = n_x + n_u
    w = [g.var('x', i) for i in range(n_x)] + [g.var('u', i) for i in range(n_u)]
    J = [[g.diff(f[m], w[j]) for j in range(nz)] for m in range(n_x)]
    kb = [g._mk('kb', value=m) for m in range(n_x)]                      # adjoint weights: plain leaves (derivative 0)
    H = []
    for i in range(nz):
        for j in range(i + 1):
            acc = g.const(0.0)
            for m in range(n_x):
                acc = g.add(acc, g.mul(kb[m], g.diff(J[m][i], w[j])))
            H.append(acc)
    return g, f, J, H, kb


def sym_source(struct_name, n_x, n_u, ode, meas=None):
    """`template <> struct ModelSym<struct_name>` with jx and jh for the right-hand side `ode` (expressions)."""
    g, f, J, H, kb = derivative_dag(n_x, n_u, ode)
    nz = n_x + n_u
    emit = g.emit
    lines1, ref1 = emit([J[m][j] for m in range(n_x) for j in range(n_x)])
    body1 = lines1 + [f"    fx[{m * n_x + j}] = {ref1[J[m][j]]};" for m in range(n_x) for j in range(n_x)]
    outs = [J[m][j] for m in range(n_x) for j in range(nz)] + H
    lines2, ref2 = emit(outs)
    body2 = (lines2 + [f"    J[{m * nz + j}] = {ref2[J[m][j]]};" for m in range(n_x) for j in range(nz)] +
             [f"    H[{q}] = {ref2[h]};" for q, h in enumerate(H)])
    nops = len(lines2)
    # structural non-zeros (bit q of JMASK: J[q], of HMASK: H[q]): the derivative phase skips the products with literal zeros - the
    # compiler may not (0 * x is not 0 for a non-finite x); 0 = no information (more than 64 entries)
    zero = ('0.0', '(-0.0)')
    jmask = sum(1 << (m * nz + j) for m in range(n_x) for j in range(nz) if ref2[J[m][j]] not in zero) if n_x * nz <= 64 else 0
    hmask = sum(1 << q for q, h in enumerate(H) if ref2[h] not in zero) if len(H) <= 64 else 0
    xmask = sum(1 << (m * n_x + j) for m in range(n_x) for j in range(n_x) if ref1[J[m][j]] not in zero) if n_x * n_x <= 64 else 0
    masks = (f"  static constexpr bool HAS_MASKS = {'true' if n_x * nz <= 64 and len(H) <= 64 else 'false'};\n"
             f"  static constexpr unsigned long long XMASK = 0x{xmask:x}ull, JMASK = 0x{jmask:x}ull, HMASK = 0x{hmask:x}ull;\n")
    meas_fn = ""
    if meas:
        # measurement map: Jy = dh/dx [NY][NX], Hy = sum_a kb[a] d2 h_a / dx2 (packed lower triangle) - the measurement term of
        # the moving-horizon estimator's cost (csrc/hilo_ocp.h::eval_derivs_sym_mhe)
        gm = Dag()
        mm = {}
        hn = [gm.from_expr(e, mm) for e in meas]
        xs = [gm.var('x', i) for i in range(n_x)]
        Jy = [[gm.diff(hn[a], xs[j]) for j in range(n_x)] for a in range(len(meas))]
        kbm = [gm._mk('kb', value=a) for a in range(len(meas))]
        Hy = []
        for i in range(n_x):
            for j in range(i + 1):
                acc = gm.const(0.0)
                for a in range(len(meas)):
                    acc = gm.add(acc, gm.mul(kbm[a], gm.diff(Jy[a][i], xs[j])))
                Hy.append(acc)
        lines3, ref3 = gm.emit(hn + [Jy[a][j] for a in range(len(meas)) for j in range(n_x)] + Hy)
        body3 = (lines3 + [f"    y[{a}] = {ref3[hn[a]]};" for a in range(len(meas))] +
                 [f"    Jy[{a * n_x + j}] = {ref3[Jy[a][j]]};" for a in range(len(meas)) for j in range(n_x)] +
                 [f"    Hy[{q}] = {ref3[h]};" for q, h in enumerate(Hy)])
        meas_fn = (f"  static constexpr bool HAS_MEAS = true;\n"
                   f"  // {len(lines3)} operations: y = h(x, u, p), Jy = dh/dx row-major [NY][NX], Hy = sum_a kb[a] d2h_a/dx2 (packed lower)\n"
                   f"  template <class P>\n"
                   f"  __device__ __forceinline__ static void mjh(const double* x, const double* u, const P* p, const double* kb,\n"
                   f"                                             double* y, double* Jy, double* Hy) {{\n"
                   f"    (void)x; (void)u; (void)p; (void)kb; (void)y; (void)Jy; (void)Hy;\n" + '\n'.join(body3) + "\n  }\n")
    else:
        meas_fn = "  static constexpr bool HAS_MEAS = false;\n"
    return (f"template <> struct ModelSym<{struct_name}> {{\n"
            f"  static constexpr bool value = true;\n" + masks +
            f"  // {len(lines1)} operations\n"
            f"  template <class P>\n"
            f"  __device__ __forceinline__ static void jx(const double* x, const double* u, const P* p, double* fx) {{\n"
            f"    (void)x; (void)u; (void)p;\n" + '\n'.join(body1) + "\n  }\n"
            f"  // {nops} operations: J = df/d(x,u) row-major [NX][NX+NU], H = sum_m kb[m] d2f_m/d(x,u)2 (packed lower triangle)\n"
            f"  template <class P>\n"
            f"  __device__ __forceinline__ static void jh(const double* x, const double* u, const P* p, const double* kb, double* J,\n"
            f"                                            double* H) {{\n"
            f"    (void)x; (void)u; (void)p; (void)kb;\n" + '\n'.join(body2) + "\n  }\n" + meas_fn + "};\n")
