"""Structural sparsity of the stage Hessians of an NMPC problem written as expressions.

The general run-time compiled policy (csrc/hilo_nmpc_user.h) obtains the Hessian of the Lagrangian of a shooting interval from
second-order Taylor sweeps, one per direction e_i and e_i + e_j (csrc/hilo_ocp.h::eval_derivs_body) - n_z (n_z + 1) / 2 sweeps
through the Runge-Kutta map per interval.  CasADi, which the reference hands its graphs to (mpc.py:1778-1787), only ever
computes the structurally non-zero entries of that Hessian; the counterpart here is a conservative pattern computed from the
expressions at `setup()`: a pair direction e_i + e_j is swept only if d2 L / dz_i dz_j can be non-zero (BASELINE configuration 5,
the mobile robot with a path variable and a soft constraint: 19 sweeps per interval instead of 66).

    pattern[a][b] = 1  iff  some term of the interval's Lagrangian can have a non-zero second derivative w.r.t. (z_a, z_b)

in the order of the augmented model z = [x (n_x), theta (n_th) | u (n_u), u_theta (n_th)].  Rules:
  * every expression (right-hand sides, generic stage cost, constraint and path expressions) contributes its own pairs
    (hilo_mpc_amd/expr.py::hessian_structure), the quadratic costs the non-zero off-diagonal entries of their weights;
  * the shooting map composes the right-hand side with itself (Runge-Kutta stages / collocation points), and with the
    continuous objective or a terminal constraint on the integrated state other terms are composed with it as well: a pair (j, l)
    of an inner function becomes every (a, b) such that j is reachable from a and l from b in the dependency graph of the
    states (a variable reaches itself) - first-order sensitivities are zero outside that reach, by induction over the stages.
Everything the analysis cannot see (zoo functors without an expression form, algebraic states) yields None = dense.
"""
import numpy as np

from .expr import Expr, hessian_structure


def _key_index(key, nx, nth, nu):
    kind, i = key
    if kind == 'x':
        return i if i < nx else None
    if kind == 'theta':
        return nx + i if i < nth else None
    if kind == 'u':
        return nx + nth + i if i < nu else None
    return None                                    # parameters and the time variable are data


def stage_hessian_pattern(ode, nx, nu, nth=0, discrete=False, Wz=None, Wdu=None, exprs=(), path_terms=(), path_weights=None,
                          composed=False, exprs_composed=()):
    """ode: the n_x right-hand sides (expressions on 'x', 'u', 'p' leaves) or None when unknown -> None (dense).
    Wz: [mza x mza] weights of the quadratic stage cost on the augmented z; Wdu: [nu x nu] input-change weights;
    exprs: further expressions of (x, u, theta): generic stage cost, stage and terminal constraint functions;
    path_terms: [(state index, reference expression of theta)], path_weights: their weight matrix (off-diagonals couple terms).
    composed: the cost terms are evaluated along the shooting map (continuous objective: at the Runge-Kutta stage points /
    collocation states) instead of at the interval's node - their pairs are lifted through the reach like the model's;
    exprs_composed: expressions that are always evaluated on the integrated state (hard terminal constraints, mpc.py:1693-1700).
    Returns a symmetric uint8 matrix [mza x mza] with a unit diagonal."""
    if ode is None:
        return None
    nxa, mza = nx + nth, nx + nth + nu + nth
    idx = lambda key: _key_index(key, nx, nth, nu)          # noqa: E731
    pairs, direct = set(), set()                             # pairs lifted through the map / taken as they are
    reach_from = [set([a]) for a in range(mza)]              # direct successors: variable a -> state s if f_s depends on a
    succ = [set() for _ in range(mza)]
    for s, e in enumerate(ode):
        dep, prs = hessian_structure(e)
        if any(k[0] in ('z', 'gp') for k in dep):
            return None
        for k in dep:
            a = idx(k)
            if a is not None:
                succ[a].add(s)
        for ka, kb in prs:
            a, b = idx(ka), idx(kb)
            if a is not None and b is not None:
                pairs.add((min(a, b), max(a, b)))
    for t in range(nth):                                     # theta' = u_theta (mpc.py:1181-1191)
        succ[nxa + nu + t].add(nx + t)
    for a in range(mza):                                     # transitive closure through the states
        todo = list(succ[a])
        while todo:
            s = todo.pop()
            if s not in reach_from[a]:
                reach_from[a].add(s)
                todo.extend(succ[s])
    cost_pairs = pairs if composed else direct
    for group, target in ((exprs, cost_pairs), (exprs_composed, pairs)):
        for e in group:
            if e is None:
                continue
            dep, prs = hessian_structure(Expr.wrap(e))
            if any(k[0] in ('z',) for k in dep):
                return None
            for ka, kb in prs:
                a, b = idx(ka), idx(kb)
                if a is not None and b is not None:
                    target.add((min(a, b), max(a, b)))
    if Wz is not None:
        W = np.asarray(Wz, dtype=float).reshape(mza, mza)
        for a in range(mza):
            for b in range(a, mza):                          # the diagonal too: composed with the map it couples what reaches a
                if W[a, b] + W[b, a] != 0.0:
                    cost_pairs.add((a, b))
    if Wdu is not None:
        W = np.asarray(Wdu, dtype=float).reshape(nu, nu)
        for a in range(nu):
            for b in range(a, nu):
                if W[a, b] + W[b, a] != 0.0:
                    cost_pairs.add((nxa + a, nxa + b))
    groups = []
    for si, r in path_terms:                                 # (x[si] - r(theta))^2
        dep, _ = hessian_structure(Expr.wrap(r))
        g = {int(si)} | {idx(k) for k in dep if idx(k) is not None}
        groups.append(g)
    if groups:
        Wp = None if path_weights is None else np.asarray(path_weights, dtype=float).reshape(len(groups), len(groups))
        coupled = Wp is not None and np.any((Wp + Wp.T)[~np.eye(len(groups), dtype=bool)] != 0.0)
        merged = [set().union(*groups)] if coupled else groups
        for g in merged:
            for a in g:
                for b in g:
                    if a <= b:
                        cost_pairs.add((a, b))
    # lift every inner pair through the reach of the variables (composition with the shooting map)
    P = np.eye(mza, dtype=np.uint8)
    R = np.zeros((mza, mza), dtype=bool)
    for a in range(mza):
        R[a, list(reach_from[a])] = True
    for j, l in pairs:
        A, B = R[:, j], R[:, l]                              # variables that reach j / l
        M = np.outer(A, B) | np.outer(B, A)
        P[M] = 1
    for a, b in direct:
        P[a, b] = P[b, a] = 1
    return P
