"""Custom constraint functions over the whole decision vector (`NMPC.set_custom_constraints_function`, optimizer.py:1180-1208;
rows `lb <= fun(v, x_ind, u_ind) <= ub` appended at the end of g, mpc.py:1729-1745).

A function of the WHOLE vector couples the stages, which a stage-wise (Riccati) solver does not see - unless the function is a SUM
OVER THE STAGES of terms that each involve the variables of one stage only,

    c_r(v) = const_r + sum_{k = 0..N} sum_j  a[r][k][j] psi_j(x_k, u_k),

which is what such functions are in practice (the reference's own use, tests/test_NMPC.py:519-552: the trapezoid integral of a
state over the horizon; also energy / resource budgets, average-value limits).  Then the row is carried by an ACCUMULATOR state of the
engine, q_{r,k+1} = q_{r,k} + sum_j a[r][k][j] psi_j(x_k, u_k), q_{r,0} = 0, and becomes a terminal row
lb <= const_r + q_{r,N} + sum_j a[r][N][j] psi_j(x_N) <= ub of the stage-structured problem - the SAME nonlinear program, with the
same KKT points, as the reference's dense row (csrc/hilo_nmpc_user.h: accumulators behind the shared slacks).  This module does the
decomposition: the user's Python function is called once with a vector of symbols, the returned expressions are split into additive
terms, every term must live on one stage, terms that are the same expression of (x, u) up to a constant factor share one psi_j.
"""
import numpy as np

from .expr import Expr

MAX_PSI = 4      # distinct stage expressions per problem (compiled into the policy; csrc/hilo_nmpc_user.h)
MAX_ROWS = 2     # custom rows = accumulator states of the engine


class _VLeaf(Expr):
    """Entry i of the decision vector (a placeholder while the user's function is evaluated)."""
    __slots__ = ()

    def __init__(self, i):
        super().__init__('v', value=int(i), name=f'v[{i}]')


def _key(e):
    """Structural fingerprint of an expression: operator, the operator's own value (the exponent of an integer power, the index of a
    learned term, ...) and the operands - `repr` leaves the value out, x**3 and x**4 would print alike."""
    if e.op == 'const':
        return repr(float(e.value))
    if not e.args:
        return f"{e.op}:{getattr(e, 'name', '')}:{e.value!r}"
    keys = [_key(a) for a in e.args]
    if e.op in ('add', 'mul'):
        keys.sort()                        # commutative: x * u and u * x are the same stage expression
    elif e.op == 'sq':
        return 'mul[None](' + keys[0] + ',' + keys[0] + ')'      # x ** 2 (expr.py: `sq`) and x * x
    return f"{e.op}[{e.value!r}](" + ','.join(keys) + ')'


def _terms(e):
    """[(coefficient, factor or None)] with e = sum coefficient * factor (None: the constant 1)."""
    e = Expr.wrap(e)
    op = e.op
    if op == 'const':
        return [(float(e.value), None)]
    if op == 'add':
        return _terms(e.args[0]) + _terms(e.args[1])
    if op == 'sub':
        return _terms(e.args[0]) + [(-c, f) for c, f in _terms(e.args[1])]
    if op == 'neg':
        return [(-c, f) for c, f in _terms(e.args[0])]
    if op in ('mul', 'div'):
        ta, tb = _terms(e.args[0]), _terms(e.args[1])

        def times(fa, fb):                      # product of two factors (None: the constant 1)
            return fb if fa is None else (fa if fb is None else fa * fb)
        if op == 'mul' and len(ta) * len(tb) <= 64:
            # distributed: (2 x_k) u_k and x_k u_k are the SAME stage expression with different coefficients
            return [(ca * cb, times(fa, fb)) for ca, fa in ta for cb, fb in tb]
        if op == 'div' and len(tb) == 1:
            cb, fb = tb[0]
            if fb is None:
                return [(c / cb, f) for c, f in ta]
            return [(c / cb, (Expr.wrap(1.0) if f is None else f) / fb) for c, f in ta]
    return [(1.0, e)]


def decompose(fun, x_ind, u_ind, n_v, model, n_rows=None):
    """Calls `fun(v, x_ind, u_ind)` with symbols and returns (psi, coef, const):
    psi    list of expressions of the model's (scaled) states and inputs - the reference's v holds SCALED variables,
    coef   [rows][N + 1][len(psi)] coefficients,
    const  [rows] constant parts.
    Raises NotImplementedError for a function that is not a sum of single-stage terms (or needs more than MAX_PSI expressions)."""
    N = len(x_ind) - 1
    where = {}
    for k, idx in enumerate(x_ind):
        for i, j in enumerate(idx):
            where[int(j)] = ('x', k, i)
    for k, idx in enumerate(u_ind):
        for i, j in enumerate(idx):
            where[int(j)] = ('u', k, i)
    v = [_VLeaf(i) for i in range(n_v)]
    out = fun(v, x_ind, u_ind)
    rows = list(out) if isinstance(out, (list, tuple, np.ndarray)) else [out]
    if n_rows is not None and len(rows) != n_rows:
        raise ValueError(f"The custom constraint function returns {len(rows)} value(s) but {n_rows} bound(s) were given.")
    if len(rows) > MAX_ROWS:
        raise NotImplementedError(f"at most {MAX_ROWS} custom constraint rows are offloaded (got {len(rows)})")
    psi, keys = [], {}
    coef = np.zeros((len(rows), N + 1, MAX_PSI))
    const = np.zeros(len(rows))
    for r, e in enumerate(rows):
        for c, f in _terms(e):
            if f is None:
                const[r] += c
                continue
            stages, bad = set(), []
            for n in f.nodes().values():
                if n.op == 'v':
                    w = where.get(int(n.value))
                    if w is None:
                        bad.append(int(n.value))
                    else:
                        stages.add(w[1])
                elif n.op in ('x', 'u', 'z', 'theta', 'p', 't', 'dt'):
                    raise NotImplementedError("a custom constraint is a function of the decision vector v only (mpc.py:1734-1742: "
                                              "`fun(v, x_ind, u_ind)`)")
            if bad:
                raise NotImplementedError(f"the custom constraint reads v[{bad[0]}], which is neither a state nor an input of a node "
                                          f"(x_ind / u_ind): only those are offloaded")
            if len(stages) != 1:
                raise NotImplementedError("the custom constraint multiplies / composes variables of DIFFERENT stages: it is not a sum "
                                          "of single-stage terms and does not fit the stage-wise solver (see hilo_mpc_amd/custom.py)")
            k = stages.pop()

            def leaf(n):
                if n.op != 'v':
                    return None
                kind, _, i = where[int(n.value)]
                return model.x[i] if kind == 'x' else model.u[i]
            g = Expr.substitute([f], leaf)[0]
            key = _key(g)
            if key not in keys:
                if len(psi) == MAX_PSI:
                    raise NotImplementedError(f"the custom constraint needs more than {MAX_PSI} distinct stage expressions")
                keys[key] = len(psi)
                psi.append(g)
            coef[r, k, keys[key]] += c
    if any(p.depends_on('u') for p in psi):
        j_u = [j for j, p in enumerate(psi) if p.depends_on('u')]
        if np.any(coef[:, N, j_u]):
            raise ValueError("the custom constraint reads an input of stage N: there is none in v")
    return psi, coef[:, :, :max(1, len(psi))] if psi else coef[:, :, :1], const
