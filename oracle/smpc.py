"""Oracle: the deterministic surrogate of the stochastic NMPC and its transcription, on the dense interior-point solver.

TEST INFRASTRUCTURE ONLY - never imported by the product package.   PARITY UNPINNED: the reference's SMPC tests
(tests/test_SMPC.py) are smoke tests without numbers, and CasADi/IPOPT cannot be installed here.

Restated from hilo_mpc/modules/controller/mpc.py:
  * `SMPC._create_deterministic_surrogate` (:2512-2614): states [mean of x | vec(Kx) column-major], the model's inputs, its
    parameters (+ the n_u n_x entries of the ancillary gain when it is not fixed, `ca.reshape(p, n_u, n_x)` column-major);
        mean+ = f(mean, u) + B_w mu_d(mean)                                                   (:2553-2559)
        Kx+   = [jode B_w] bigK [jode B_w]^T                                                  (:2597-2599)
        bigK  = [[Kz, Kzd], [Kzd^T, Kd]],  Kz = [[Kx, Kx K^T], [K Kx, K Kx K^T]],  Kzd = Kz jgp^T,  Kd = Kd0 + jgp Kz jgp^T
    with mu_d / Kd0 the GP posterior mean / variance INCLUDING the noise variance (`gp.predict(x)` with noise_free=False,
    gp.py:699-713; inference.py:211-216) at the mean state, jgp = d mu_d / d(x, u) and jode = d f / d(x, u); the surrogate is
    discrete and set up with dt = 1 (:2483).
  * chance constraints (:2623-2645): rows  x_i + sqrt(2) erfinv(2 p_i - 1) sqrt(Kx_ii + 1e-8) <= x_ub_i  and
    -x_i + sqrt(2) erfinv(2 p_i - 1) sqrt(Kx_ii + 1e-8) <= -x_lb_i  as hard stage AND terminal constraints; box bounds of the
    surrogate (:2676-2695): the user's bounds on the mean, [0, inf) on the diagonal of Kx, nothing off it.
  * cost (:2766-2771): the quadratic terms on the mean + trace(Q Kx) + trace(R Ku), Ku = K Kx K^T, as a generic stage cost
    (evaluated on the scaled variables like every generic cost).
Here sympy plays CasADi's role: all derivatives are symbolic.
"""
from __future__ import annotations

import numpy as np
import sympy as sp
from scipy.special import erfinv

from .models import OracleModel
from .nmpc_gen import GenNmpcProblem


def gp_symbolic(post, feats):
    """(mean, variance incl. noise) of an oracle GP posterior (oracle/gp.py::Posterior, squared-exponential kernel, constant /
    zero mean) as sympy expressions of the feature symbols `feats` (inference.py:211-216 written out term by term)."""
    spec = post.kernel_spec['kwargs']
    ad = list(spec.get('active_dims', range(len(feats))))
    ls = np.broadcast_to(np.asarray(spec.get('length_scales', 1.), dtype=float), (len(ad),))
    M = np.exp(-2 * np.log(ls))                                              # kernel.py:538-555
    sf2 = float(np.exp(2 * (np.log(spec.get('signal_variance', 1.)) / 2)))   # kernel.py:127-130
    bias = 0.
    if post.mean_spec.get('type') == 'constant':
        bias = float(post.mean_spec.get('kwargs', {}).get('bias', 0.))
    elif post.mean_spec.get('type') not in ('zero', None):
        raise NotImplementedError(post.mean_spec)
    n = post.X.shape[1]
    ks = [sf2 * sp.exp(-sp.Rational(1, 2) * sum(float(M[q]) * (feats[ad[q]] - float(post.X[ad[q], i])) ** 2 for q in range(len(ad))))
          for i in range(n)]
    mean = sp.Float(bias) + sum(float(post.alpha[i]) * ks[i] for i in range(n))
    Linv = np.linalg.inv(post.R.T)                                           # v = L^-1 k*, var = k** - v^T v
    v = [sum(float(Linv[i, j]) * ks[j] for j in range(i + 1)) for i in range(n)]
    var = sp.Float(sf2) + sp.Float(post.sn2) - sum(vi ** 2 for vi in v)
    return mean, var


def smpc_surrogate(model: OracleModel, posts, features, Bw, Kgain=None):
    """model: DISCRETE OracleModel (x+ = f); posts: GP posteriors; features[k]: state indices the k-th GP reads."""
    assert model.discrete
    nx, nu = model.nx, model.nu
    Bw = np.atleast_2d(np.asarray(Bw, dtype=float)).reshape(nx, len(posts))
    kx = [sp.Symbol(f'kx_{k}') for k in range(nx * nx)]
    Kx = sp.Matrix(nx, nx, lambda i, j: kx[j * nx + i])                      # column-major
    p = list(model.p)
    if Kgain is None:
        kg = [sp.Symbol(f'kgain_{i}') for i in range(nx * nu)]
        p += kg
        K = sp.Matrix(nu, nx, lambda i, j: kg[j * nu + i])
    else:
        K = sp.Matrix(np.atleast_2d(np.asarray(Kgain, dtype=float)).reshape(nu, nx))
    w = sp.Matrix(model.x + model.u)
    f = sp.Matrix(model.ode).subs(model.dt, 1)
    mu, var = [], []
    for post, ft in zip(posts, features):
        m, v = gp_symbolic(post, [model.x[i] for i in ft])
        mu.append(m), var.append(v)
    mu = sp.Matrix(mu)
    jgp = mu.jacobian(w)
    jode = f.jacobian(w)
    Kxu = Kx * K.T
    Kz = sp.Matrix(sp.BlockMatrix([[Kx, Kxu], [Kxu.T, K * Kx * K.T]]))
    Kd = sp.diag(*var) + jgp * Kz * jgp.T
    Kzd = Kz * jgp.T
    bigK = sp.Matrix(sp.BlockMatrix([[Kz, Kzd], [Kzd.T, Kd]]))
    jB = jode.row_join(sp.Matrix(Bw))
    ode_c = jB * bigK * jB.T
    ode = list(f + sp.Matrix(Bw) * mu) + [ode_c[i, j] for j in range(nx) for i in range(nx)]
    return OracleModel(model.name + '_smpc', -1, model.x + kx, model.u, p, ode, discrete=True), Kx, K


def smpc_problem(model, posts, features, Bw, N, Kgain_value, Kgain_is_parameter=True, stage_states=None, stage_inputs=None,
                 terminal_states=None, x_lb=None, x_ub=None, x_lb_p=None, x_ub_p=None, u_lb=None, u_ub=None, **kw):
    """The NLP `SMPC.setup()` builds.  Kgain_value: the gain used in the cost trace(R Ku) (and in the model when the gain is not a
    parameter).  stage_* / terminal_states index the ORIGINAL states / inputs like `quad_stage_cost.add_states`."""
    nx, nu = model.nx, model.nu
    Kv = np.atleast_2d(np.asarray(Kgain_value, dtype=float)).reshape(nu, nx)
    sur, Kx, _ = smpc_surrogate(model, posts, features, Bw, None if Kgain_is_parameter else Kv)
    inf = np.inf
    con = tcon = None
    box = {}
    if x_lb is not None or x_ub is not None or u_lb is not None or u_ub is not None:     # set_box_chance_constraints was called
        xl = np.full(nx, -inf) if x_lb is None else np.asarray(x_lb, dtype=float)
        xu = np.full(nx, inf) if x_ub is None else np.asarray(x_ub, dtype=float)
        pl = np.full(nx, .954) if x_lb_p is None else np.asarray(x_lb_p, dtype=float)
        pu = np.full(nx, .954) if x_ub_p is None else np.asarray(x_ub_p, dtype=float)
        sd = [sp.sqrt(Kx[i, i] + 1e-8) for i in range(nx)]
        rows = [model.x[i] + float(np.sqrt(2.) * erfinv(2 * pu[i] - 1)) * sd[i] for i in range(nx)] + \
               [-model.x[i] + float(np.sqrt(2.) * erfinv(2 * pl[i] - 1)) * sd[i] for i in range(nx)]
        con = dict(expr=rows, lb=[-inf] * (2 * nx), ub=list(xu) + list(-xl))
        tcon = dict(expr=rows, lb=[-inf] * (2 * nx), ub=list(xu) + list(-xl))
        if x_ub is not None:
            box['x_ub'] = list(xu) + [inf] * (nx * nx)
        if x_lb is not None:
            box['x_lb'] = list(xl) + [0. if i == j else -inf for i in range(nx) for j in range(nx)]
        box['u_lb'], box['u_ub'] = u_lb, u_ub
    # quad_stage_cost.Q / R (modeling.py:492-512) restricted to the original states
    Q, R = np.zeros((nx, nx)), np.zeros((nu, nu))
    for ind, W, _ in (stage_states or []):
        Q[np.ix_(list(ind), list(ind))] += np.diag(np.broadcast_to(np.asarray(W, dtype=float), (len(ind),))) if np.ndim(W) < 2 else np.asarray(W)
    for ind, W, _ in (stage_inputs or []):
        R[np.ix_(list(ind), list(ind))] += np.diag(np.broadcast_to(np.asarray(W, dtype=float), (len(ind),))) if np.ndim(W) < 2 else np.asarray(W)
    Ku = sp.Matrix(Kv) * Kx * sp.Matrix(Kv).T
    gen = (sp.Matrix(Q) * Kx).trace() + (sp.Matrix(R) * Ku).trace()
    return GenNmpcProblem(sur, 1., N, constraint=con, terminal_constraint=tcon, generic_stage=gen, stage_states=stage_states,
                          stage_inputs=stage_inputs, terminal_states=terminal_states, **box, **kw)
