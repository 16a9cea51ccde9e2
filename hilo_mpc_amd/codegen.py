"""Expression trees -> HIP source of the functors the engine is compiled against at run time.

The reference hands CasADi graphs to `ca.nlpsol`, which generates and compiles derivative code at `setup()`
(hilo_mpc/modules/controller/mpc.py:1778-1787).  The counterpart here: the expressions a user writes on `model.x`,
`model.u`, `model.p` (hilo_mpc_amd/expr.py) are emitted as templated C++ in the shape of the device zoo
(hilo_mpc_amd/csrc/hilo_models.h) - the scalar type carries the derivatives (Dual<N>, Jet2, hilo_ad.h) - and compiled
with hiprtc by the library (csrc/hilo_jit.hip; `desc.user_source` of include/hilo_hip.h).

Emission rules: one `const auto tK = ...;` per distinct inner node, in the order the nodes were CREATED - the order in which
the user's Python statements were evaluated, so the emitted function reads like the statements that built it (and a zoo
functor written in the same order compiles to the same code: tests/test_jit_gpu.py compares them bit for bit); shared
sub-expressions - the same Python object used twice - are evaluated once, exactly like a named temporary of a hand-written
functor; leaves are array reads, constants are printed with `repr` (round-trip exact).
"""
from .expr import Expr

_BIN = {'add': '+', 'sub': '-', 'mul': '*', 'div': '/'}
_FUN = {n: n for n in ('sq', 'sin', 'cos', 'exp', 'log', 'sqrt', 'log10', 'fabs', 'sign', 'asin', 'acos', 'atan', 'asinh', 'acosh',
                       'atanh')}        # csrc/hilo_ad.h: one overload per scalar type


def _lit(v):
    v = float(v)
    if v != v or v in (float('inf'), float('-inf')):
        raise ValueError("non-finite constant in an expression")
    s = repr(v)
    if 'e' not in s and '.' not in s and 'n' not in s:
        s += '.0'
    return s


class Emitter:
    """Collects statements for a set of expressions that share sub-expressions."""

    def __init__(self, theta_index=None, x='x', u='u', p='p'):
        self.theta_index, self.names = theta_index, {'x': x, 'u': u, 'p': p, 'z': 'z'}
        self.lines, self.memo = [], {}

    def ref(self, e):
        """Name of the value of `e`; emits the statements of its not-yet-emitted inner nodes in creation order."""
        e = Expr.wrap(e)
        if id(e) in self.memo:
            return self.memo[id(e)]
        todo, seen, stack = [], set(), [e]
        while stack:                                  # inner nodes below e that still need a statement
            n = stack.pop()
            if id(n) in seen or id(n) in self.memo:
                continue
            seen.add(id(n))
            if n.args:
                todo.append(n)
                stack.extend(n.args)
        for n in sorted(todo, key=lambda q: q.serial):
            self._one(n)
        return self._one(e)

    def _one(self, e):
        if id(e) in self.memo:
            return self.memo[id(e)]
        op = e.op
        if op == 'const':
            r = _lit(e.value)
        elif op in ('x', 'u', 'p', 'z'):
            r = f"{self.names[op]}[{int(e.value)}]"
        elif op == 'theta':
            if self.theta_index is None:
                raise ValueError("a path variable can only appear where the path variable is defined (path references, costs)")
            r = f"{self.names['x']}[{int(self.theta_index) + int(e.value)}]"
        else:
            a = [self._one(c) for c in e.args]
            if op in _BIN:
                rhs = f"{a[0]} {_BIN[op]} {a[1]}"
            elif op == 'neg':
                rhs = f"-1.0 * ({a[0]})"
            elif op in _FUN:
                rhs = f"{_FUN[op]}({a[0]})"
            elif op == 'atan2':
                rhs = f"atan2({a[0]}, {a[1]})"
            elif op == 'gp':
                # posterior mean of learned term #value at the features a[...] (csrc/hilo_models.h::gp_se_mean); the features
                # are brought to their common scalar type (states and inputs may carry different derivative types)
                g = f"g{len(self.lines)}"
                self.lines.append(f"    using {g}_t = decltype({' + '.join(['0.0'] + a)});")
                self.lines.append(f"    const {g}_t {g}[] = {{{', '.join(f'{g}_t({q})' for q in a)}}};")
                rhs = f"gp_se_mean(hilo_user_gp[{int(e.value)}], {g})"
            elif op == 'gpk':
                # posterior mean of a learned term with a general kernel: the helper emitted by gp_helper_source (same table as
                # gp_se_mean, all features as columns)
                g = f"g{len(self.lines)}"
                self.lines.append(f"    using {g}_t = decltype({' + '.join(['0.0'] + a)});")
                self.lines.append(f"    const {g}_t {g}[] = {{{', '.join(f'{g}_t({q})' for q in a)}}};")
                rhs = f"hilo_user_gpk{int(e.value)}(hilo_user_gp[{int(e.value)}], {g})"
            elif op in ('gpvar', 'gpd'):
                # posterior variance of a learned term (with the noise variance, `gp.predict(x)[1]`, gp.py:699-713) and the
                # derivative of its posterior mean with respect to feature j: the terms of the covariance propagation of the
                # stochastic NMPC (mpc.py:2527-2575); csrc/hilo_models.h::gp_se_var / gp_se_dmean
                g = f"g{len(self.lines)}"
                self.lines.append(f"    using {g}_t = decltype({' + '.join(['0.0'] + a)});")
                self.lines.append(f"    const {g}_t {g}[] = {{{', '.join(f'{g}_t({q})' for q in a)}}};")
                rhs = (f"gp_se_var(hilo_user_gp[{int(e.value)}], {g})" if op == 'gpvar' else
                       f"gp_se_dmean(hilo_user_gp[{int(e.value[0])}], {g}, {int(e.value[1])})")
            elif op == 'powi':
                n = int(e.value)
                if n == 0:
                    rhs = "1.0"
                else:
                    prod = ' * '.join([a[0]] * abs(n))
                    rhs = prod if n > 0 else f"1.0 / ({prod})"
            else:
                raise ValueError(f"cannot emit operator '{op}'")
            r = f"t{len(self.lines)}"
            self.lines.append(f"    const auto {r} = {rhs};")
        self.memo[id(e)] = r
        return r


def _fn(ret, name, args, body_lines, result_lines):
    return (f"  template <class T>\n  __device__ __forceinline__ static {ret} {name}({args}) {{\n" +
            '\n'.join(body_lines + result_lines) + "\n  }\n")


def gp_helper_source(k, nf, kexpr):
    """Posterior mean of learned term #k with the kernel `kexpr` (an expression of the features, leaves ('x', q), and of a
    training point, leaves ('p', q); hilo_mpc_amd/gp.py::kernel_expr) in the scalar type T of the caller: bias + sum_i alpha_i
    k(f, X_i) over the table [n, nf, -, bias, (nf + nf entries), rows (X_0..X_{nf-1}, alpha)] of csrc/hilo_gp.hip::gp_pack_se."""
    em = Emitter(x='f', p='r')
    ref = em.ref(kexpr)
    body = '\n'.join('  ' + ln for ln in em.lines)
    return (f"template <class T>\n__device__ __forceinline__ T hilo_user_gpk{k}(const double* g, const T* f) {{\n"
            f"  const int n = (int)g[0];\n  const double* r = g + {4 + 2 * nf};\n  T acc = T(0.0);\n"
            f"  for (int i = 0; i < n; ++i, r += {nf + 1}) {{\n{body}\n      acc = acc + r[{nf}] * ({ref});\n  }}\n"
            f"  return g[3] + acc;\n}}\n")


def model_source(n_x, n_u, n_p, ode, meas, discrete, helpers=()):
    """`struct UserModel` for the right-hand side `ode` (list of n_x expressions) and the measurement map `meas`; `helpers`:
    source of the functions the expressions call (learned terms with general kernels)."""
    if len(ode) != n_x:
        raise ValueError(f"the model has {n_x} states but {len(ode)} dynamical equations")
    em = Emitter()
    dx = [em.ref(e) for e in ode]
    body = em.lines + [f"    dx[{i}] = T({r});" for i, r in enumerate(dx)]
    em2 = Emitter()
    yy = [em2.ref(e) for e in meas]
    body2 = em2.lines + [f"    y[{i}] = T({r});" for i, r in enumerate(yy)]
    n_y = len(meas)
    src = (''.join(helpers) +
           f"struct UserModel {{\n"
           f"  static constexpr int NX = {n_x}, NU = {n_u}, NP = {n_p}, NY = {n_y};\n"
           f"  static constexpr bool DISCRETE = {'true' if discrete else 'false'};\n"
           f"  template <class T, class U, class P>\n"
           f"  __device__ __forceinline__ static void ode(const T* x, const U* u, const P* p, double dt, T* dx) {{\n"
           f"    (void)x; (void)u; (void)p; (void)dt;\n" + '\n'.join(body) + "\n  }\n"
           f"  template <class T, class U, class P>\n"
           f"  __device__ __forceinline__ static void meas(const T* x, const U* u, const P* p, double dt, T* y) {{\n"
           f"    (void)x; (void)u; (void)p; (void)dt; (void)y;\n" + '\n'.join(body2) + "\n  }\n};\n")
    # symbolic first / second derivatives for the engine's derivative phase (csrc/hilo_ocp.h::eval_derivs_sym); a model with
    # a learned term keeps the Taylor sweeps
    learned = ('gp', 'gpvar', 'gpd', 'gpk')
    if not any(n.op in learned for e in ode for n in Expr.wrap(e).nodes().values()):
        from .symdiff import sym_source
        src += sym_source('UserModel', n_x, n_u, ode,
                          meas if not any(n.op in learned for e in meas for n in Expr.wrap(e).nodes().values()) else None)
    return src


def dae_model_source(n_x, n_u, n_p, n_z, ode, alg, meas, z_guess, alg_at_slope=False):
    """`struct UserModel` of a semi-explicit index-1 DAE  dx/dt = f(x, z, u, p),  0 = g(x, z, u, p).
    alg_at_slope: the reference's EXPLICIT Runge-Kutta transcription hands the algebraic equations the stage's SLOPE where the state
    belongs (`alg(t + h c_i, dk, Z[:, i], u, p)`, hilo_mpc/util/modeling.py:1268): the stage's algebraic variables solve
    g(f(X_i, Z_i, u), Z_i, u) = 0.  Restated as it is - the emitted `alg` (and its dg/dz) is that composition, everything else the same.  The engine sees the ODE
    dx/dt = f(x, zeta(x, u, p), u, p): `ode` solves the algebraic equations for z by Newton's method IN THE SCALAR TYPE it is
    called with (values, forward duals, second-order Taylor numbers - each sweep after the values have converged fixes one more
    derivative order; csrc/hilo_models.h::dae_solve), so the derivatives the interior point needs are those of the implicit
    function.  `ode_z`, `alg`, `alg_jz` (dg/dz, symbolic) are the raw functions; the output pass of the collocation policy
    reconstructs z at the collocation points and the multipliers of the algebraic rows from them."""
    from .symdiff import Dag
    if len(ode) != n_x or len(alg) != n_z:
        raise ValueError("dimension mismatch between states and equations")
    if any(n.op in ('gp', 'gpk') for e in list(ode) + list(alg) for n in Expr.wrap(e).nodes().values()):
        raise NotImplementedError("a learned term inside a DAE model is not built")
    if alg_at_slope:
        odes = [Expr.wrap(e) for e in ode]
        alg = Expr.substitute([Expr.wrap(e) for e in alg], lambda n: odes[int(n.value)] if n.op == 'x' else None)
    em = Emitter()
    dx = [em.ref(e) for e in ode]
    body_f = em.lines + [f"    dx[{i}] = T({r});" for i, r in enumerate(dx)]
    em = Emitter()
    rr = [em.ref(e) for e in alg]
    body_g = em.lines + [f"    r[{i}] = T({q});" for i, q in enumerate(rr)]
    g = Dag()
    memo = {}
    gn = [g.from_expr(e, memo) for e in alg]
    zs = [g.var('z', i) for i in range(n_z)]
    Jz = [g.diff(gn[a], zs[b]) for a in range(n_z) for b in range(n_z)]
    lines, ref = g.emit(Jz, generic=True)
    body_j = lines + [f"    J[{q}] = T({ref[n]});" for q, n in enumerate(Jz)]
    em = Emitter()
    yy = [em.ref(e) for e in meas]
    body_y = em.lines + [f"    y[{i}] = T({r});" for i, r in enumerate(yy)]
    zg = ', '.join(_lit(v) for v in z_guess)
    sig = "const T* x, const T* z, const T* u, const P* p"
    return (f"struct UserModel {{\n"
            f"  static constexpr int NX = {n_x}, NU = {n_u}, NP = {n_p}, NY = {len(meas)}, NZ = {n_z};\n"
            f"  static constexpr bool DISCRETE = false;\n"
            f"  __device__ __forceinline__ static double z_guess(int i) {{ const double g[{n_z}] = {{{zg}}}; return g[i]; }}\n"
            f"  template <class T, class P>\n  __device__ __forceinline__ static void ode_z({sig}, T* dx) {{\n"
            f"    (void)x; (void)z; (void)u; (void)p;\n" + '\n'.join(body_f) + "\n  }\n"
            f"  template <class T, class P>\n  __device__ __forceinline__ static void alg({sig}, T* r) {{\n"
            f"    (void)x; (void)z; (void)u; (void)p;\n" + '\n'.join(body_g) + "\n  }\n"
            f"  template <class T, class P>\n  __device__ __forceinline__ static void alg_jz({sig}, T* J) {{\n"
            f"    (void)x; (void)z; (void)u; (void)p;\n" + '\n'.join(body_j) + "\n  }\n"
            f"  template <class T, class P>\n  __device__ __forceinline__ static void meas_z({sig}, T* y) {{\n"
            f"    (void)x; (void)z; (void)u; (void)p; (void)y;\n" + '\n'.join(body_y) + "\n  }\n"
            f"  static constexpr bool ODE_USES_Z = {'true' if any(Expr.wrap(e).depends_on('z') for e in ode) else 'false'}, "
            f"MEAS_USES_Z = {'true' if any(Expr.wrap(e).depends_on('z') for e in meas) else 'false'};\n"
            f"  template <class T, class U, class P>\n"
            f"  __device__ __forceinline__ static void ode(const T* x, const U* u, const P* p, double, T* dx) {{\n"
            f"    dae_ode<UserModel>(x, u, p, dx);\n  }}\n"
            f"  template <class T, class U, class P>\n"
            f"  __device__ __forceinline__ static void meas(const T* x, const U* u, const P* p, double, T* y) {{\n"
            f"    dae_meas<UserModel>(x, u, p, y);\n  }}\n}};\n")


def zoo_alias(functor):
    return f"using UserModel = {functor};\n"


def fun_source(n_x, stage=None, term=None, con=(), tcon=(), path_stage=(), path_term=(), acc=()):
    """`struct UserFun` (csrc/hilo_nmpc_user.h): generic costs (scaled variables), constraint expressions (un-scaled
    variables; `z`: the algebraic states of a DAE model at the point, NULL when no expression names one), path references
    (functions of the path variable = state index n_x)."""
    uses_z = any(Expr.wrap(e).depends_on('z') for e in con)
    for what, ee in (('cost', [e for e in (stage, term) if e is not None]), ('terminal constraint', tcon),
                     ('path reference', list(path_stage) + list(path_term)), ('custom constraint', acc)):
        if any(Expr.wrap(e).depends_on('z') for e in ee):
            raise NotImplementedError(f"an algebraic state inside a {what} is not offloaded (stage constraints may name them)")
    s = ("struct UserFun {\n"
         f"  static constexpr bool HAS_STAGE = {'true' if stage is not None else 'false'}, "
         f"HAS_TERM = {'true' if term is not None else 'false'};\n"
         f"  static constexpr int NEXPR = {len(con)}, NTEXPR = {len(tcon)}, NPS = {len(path_stage)}, NPT = {len(path_term)}, "
         f"NACC = {len(acc)};\n"
         f"  static constexpr bool CON_USES_Z = {'true' if uses_z else 'false'};\n")
    if stage is not None:
        em = Emitter(theta_index=n_x)
        r = em.ref(stage)
        s += _fn('T', 'stage', 'const T* x, const T* u, const double* p', ["    (void)x; (void)u; (void)p;"] + em.lines,
                 [f"    return T({r});"])
    if term is not None:
        if Expr.wrap(term).depends_on('u'):
            raise ValueError("The terminal cost can only contain states")
        em = Emitter(theta_index=n_x)
        r = em.ref(term)
        s += _fn('T', 'term', 'const T* x, const double* p', ["    (void)x; (void)p;"] + em.lines, [f"    return T({r});"])
    for name, exprs, args in (('con', con, 'const T* x, const T* u, const T* z, const double* p, T* c'),
                              ('tcon', tcon, 'const T* x, const T* u, const double* p, T* c')):
        if exprs:
            em = Emitter(theta_index=n_x)      # (x is the augmented state: a constraint may involve the path variable)
            rr = [em.ref(e) for e in exprs]
            s += _fn('void', name, args, ["    (void)x; (void)u; (void)p;" + (" (void)z;" if name == 'con' else "")] + em.lines,
                     [f"    c[{i}] = T({r});" for i, r in enumerate(rr)])
    if acc:      # stage expressions of a custom constraint function (hilo_mpc_amd/custom.py): SCALED states / inputs, like `stage`
        em = Emitter(theta_index=n_x)
        rr = [em.ref(e) for e in acc]
        s += _fn('void', 'acc', 'const T* x, const T* u, const double* p, T* c', ["    (void)x; (void)u; (void)p;"] + em.lines,
                 [f"    c[{i}] = T({r});" for i, r in enumerate(rr)])
    for name, exprs in (('path_stage', path_stage), ('path_term', path_term)):
        if exprs:
            em = Emitter(theta_index=n_x)
            rr = [em.ref(e) for e in exprs]
            s += _fn('void', name, 'const T* x, const double* p, T* r', ["    (void)x; (void)p;"] + em.lines,
                     [f"    r[{i}] = T({q});" for i, q in enumerate(rr)])
    return s + "};\n"


def mhe_fun_source(con):
    """`struct UserFun` of an estimator's stage constraint (csrc/hilo_mhe_policy.h::MheGen): the expressions of the (scaled) states
    and parameters, mhe.py:498-508."""
    for e in con:
        e = Expr.wrap(e)
        if e.depends_on('u') or e.depends_on('z') or e.depends_on('theta'):
            raise ValueError("the estimator's stage constraint is a function of the states and parameters (mhe.py:501-508: the inputs "
                             "are data)")
    em = Emitter()
    rr = [em.ref(e) for e in con]
    return ("struct UserFun {\n"
            f"  static constexpr int NEXPR = {len(con)};\n" +
            _fn('void', 'con', 'const T* x, const T* p, T* c', ["    (void)x; (void)p;"] + em.lines,
                [f"    c[{i}] = T({r});" for i, r in enumerate(rr)]) + "};\n")
