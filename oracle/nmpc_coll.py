"""Oracle: NMPC with the reference's DEFAULT transcription - direct collocation (Lagrange basis at Radau points).

TEST INFRASTRUCTURE ONLY - never imported by the product package.

PINNED by the reference's only published NMPC numbers: docs/docsource/examples/CSTR_Example.ipynb (cells 4/6/14/16) prints
'True: Q: 59882.1817 C_A: 0.4912 C_B: 0.5088 T: 438.4732' after 1000 closed-loop steps of the economic NMPC (GenericCost,
default collocation Radau-3, continuous objective); this module + the interior-point method of oracle/nmpc.py print the same
line (tests/golden/make_cstr_golden.py -> tests/golden/nmpc_cstr.json, tests/test_oracle_nmpc_coll.py).

Restated from hilo_mpc/util/modeling.py:1091-1211 (`RungeKutta._construct_polynomial_basis`, `_collocation`) and
hilo_mpc/modules/controller/mpc.py:1307-1372, :1497-1518, :1657-1666 for a CONTINUOUS model with
`integration_method='collocation'` (optimizer.py:1410-1418 defaults: 'radau', degree 3) and the default discrete objective:
  tau = [0] + collocation_points(d, 'radau')          (modeling.py:1108; Radau: roots of P_{d-1} - P_d mapped to (0, 1])
  Lagrange basis L_i on tau:  D_i = L_i(1), C[i, j] = L_i'(tau_j), B_i = int_0^1 L_i       (modeling.py:1110-1124)
  per interval, collocation states x_{k,1..d} are decision variables (mpc.py:1501-1509; bounds / guess = the state's, tiled,
  :1321-1323) with the equations  dt f(x_{k,i}, u_k) - sum_j C[j, i] x_{k,j} = 0,  x_{k,0} = x_k   (modeling.py:1183-1189)
  and the continuity row  x_{k+1} - sum_j D_j x_{k,j} = 0                                    (modeling.py:1180,1192; mpc.py:1667)
  v = [x_0..x_N | u_0..u_{N-1} | ip_0..ip_{N-1}],  g per stage = [collocation rows (d nx) | continuity (nx)]  (mpc.py:1657-1669)
  objective: the discrete sum of the stage cost at the shooting nodes + terminal cost (mpc.py:1676-1682).
The model is the scaled one (hilo_mpc/modules/base.py:1562-1591).
"""
from __future__ import annotations

import numpy as np
from numpy.polynomial import legendre

from .nmpc import DenseIpm, IpmOptions, NmpcProblem

INF = np.inf


def collocation_points(d, method='radau'):
    """`casadi.collocation_points(d, method)` restated: Gauss-Radau (right end point included) or Gauss-Legendre on (0, 1]."""
    if method == 'radau':
        c = np.zeros(d + 1)
        c[d - 1], c[d] = 1., -1.
        r = np.sort(np.real(legendre.legroots(c)))
    elif method == 'legendre':
        c = np.zeros(d + 1)
        c[d] = 1.
        r = np.sort(np.real(legendre.legroots(c)))
    else:
        raise ValueError(method)
    return list((r + 1.) / 2.)


def polynomial_basis(d, method='radau'):
    """B, C, D, tau of modeling.py:1091-1127."""
    tau = [0.] + collocation_points(d, method)
    B, C, D = np.zeros(d + 1), np.zeros((d + 1, d + 1)), np.zeros(d + 1)
    for i in range(d + 1):
        L = np.poly1d([1.])
        for j in range(d + 1):
            if j != i:
                L *= np.poly1d([1., -tau[j]]) / (tau[i] - tau[j])
        D[i] = L(1.)
        Ld = np.polyder(L)
        for j in range(d + 1):
            C[i, j] = Ld(tau[j])
        B[i] = np.polyint(L)(1.)
    return B, C, D, np.array(tau)


class CollNmpcProblem(NmpcProblem):
    """`objective`: 'discrete' = sum_k l(x_k, u_k) (mpc.py:1676-1680), 'continuous' = the quadrature of the Lagrange term
    through the collocation polynomial, sum_k dt sum_i B_i l(x_{k,i}, u_k) (modeling.py:1195; the reference's default for a
    continuous model, optimizer.py:1423-1426).
    `generic_stage` / `generic_term`: sympy expressions of the model's symbols, the `GenericCost` of
    `nmpc.stage_cost.cost = ...` / `nmpc.terminal_cost.cost = ...` (modeling.py:38-87, mpc.py:213-218).  QUIRK restated: the
    expression is attached to the model as its quadrature function AFTER the model was scaled (mpc.py:1210 then :1283), and
    `Model.scale` only substitutes inside the model's own equations (base.py:1169-1179), so the cost's symbols ARE the scaled
    NLP variables: l is evaluated at (x / x_scaling, u / u_scaling), unlike the dynamics."""

    def __init__(self, model, dt, N, degree=3, points='radau', objective='discrete', generic_stage=None,
                 generic_term=None, **kw):
        super().__init__(model, dt, N, **kw)
        assert not model.discrete, "collocation needs the continuous model"
        assert objective in ('discrete', 'continuous')
        self.objective = objective
        self.gen_stage = self._gen(generic_stage, model.x + model.u)
        self.gen_term = self._gen(generic_term, model.x)
        self.d = degree
        self.B, self.C, self.D, self.tau = polynomial_basis(degree, points)
        nx, nu, d = self.nx, self.nu, degree
        off = (N + 1) * nx + N * nu
        self.ip_ind = [list(range(off + k * d * nx, off + (k + 1) * d * nx)) for k in range(N)]   # mpc.py:1501-1509
        self.n_v = off + N * d * nx                                                               # mpc.py:1440-1443
        self.n_g = N * (d * nx + nx)
        self.v_lb = np.concatenate([self.v_lb, np.tile(self.x_lb, N * d)])
        self.v_ub = np.concatenate([self.v_ub, np.tile(self.x_ub, N * d)])
        self.v_guess = np.concatenate([self.v_guess, np.tile(self.x_guess, N * d)])

    def _gen(self, expr, w):
        """value / gradient / Hessian callables of a generic cost w.r.t. the symbols `w` (scaled variables, see class doc)."""
        if expr is None:
            return None
        import sympy as sp
        from .models import _lam
        m = self.model
        e = sp.sympify(expr)
        args = [m.x, m.u, m.p]
        g = [sp.diff(e, a) for a in w]
        H = [[sp.diff(e, a, b) for b in w] for a in w]
        return _lam([e], args), _lam(g, args), _lam(H, args)

    def lagrange(self, xs, us, p, k, u_old, need=0):
        """Lagrange term l(x, u) of interval k on scaled variables: quadratic part (+ input change in interval 0) + generic
        part.  Returns value [B] (and gradient [B,nz], Hessian [B,nz,nz] w.r.t. z = (xs, us) when need > 0)."""
        nx = self.nx
        z = np.concatenate([xs, us], axis=1) - self.zref
        B = z.shape[0]
        f = np.einsum('bi,ij,bj->b', z, self.Wz, z)
        if need:
            g = z @ (self.Wz + self.Wz.T)
            H = np.broadcast_to(self.Wz + self.Wz.T, (B, self.nz, self.nz)).copy()
        if k == 0 and u_old is not None:
            dd = us - u_old
            f = f + np.einsum('bi,ij,bj->b', dd, self.Wdu, dd)
            if need:
                g[:, nx:] += dd @ (self.Wdu + self.Wdu.T)
                H[:, nx:, nx:] += self.Wdu + self.Wdu.T
        if self.gen_stage is not None:
            pv = p if self.np_ else np.zeros((B, 0))
            f = f + self.gen_stage[0](xs, us, pv)[:, 0]
            if need:
                g += self.gen_stage[1](xs, us, pv)
                H += self.gen_stage[2](xs, us, pv)
        return (f, g, H) if need else f

    def mayer(self, xs, p, need=0):
        d = xs - self.xrefN
        B = xs.shape[0]
        f = np.einsum('bi,ij,bj->b', d, self.WN, d)
        if need:
            g = d @ (self.WN + self.WN.T)
            H = np.broadcast_to(self.WN + self.WN.T, (B, self.nx, self.nx)).copy()
        if self.gen_term is not None:
            pv = p if self.np_ else np.zeros((B, 0))
            u0 = np.zeros((B, self.nu))
            f = f + self.gen_term[0](xs, u0, pv)[:, 0]
            if need:
                g += self.gen_term[1](xs, u0, pv)
                H += self.gen_term[2](xs, u0, pv)
        return (f, g, H) if need else f

    # scaled continuous right-hand side with first / second derivatives w.r.t. (xs, us)
    def rhs(self, xs, us, p, need=0):
        x, u = xs * self.sx, us * self.su
        sm = self.smap
        B = x.shape[0]
        u = np.broadcast_to(u, (B, self.nu))
        p = np.broadcast_to(np.atleast_2d(p), (B, self.np_)) if self.np_ else np.zeros((B, 0))
        if need == 0:
            return sm._f(x, u, p, self.dt) / self.sx
        f, fw, fww = sm._rhs(x, u, p, self.dt)
        sz = np.concatenate([self.sx, self.su])
        fw = fw * sz[None, None, :] / self.sx[None, :, None]
        fww = fww * sz[None, None, :, None] * sz[None, None, None, :] / self.sx[None, :, None, None]
        return f / self.sx, fw, fww


class CollIpm(DenseIpm):
    """Free variables w = [x_1..x_N | u_0..u_{N-1} | X_0..X_{N-1}]  (X_k = the d collocation states of interval k)."""

    def __init__(self, prob: CollNmpcProblem, options: IpmOptions | None = None):
        self.pb = pb = prob
        self.o = o = options or IpmOptions()
        N, nx, nu, d = pb.N, pb.nx, pb.nu, pb.d
        self.o_u = N * nx
        self.o_c = self.o_u + N * nu
        self.nw = self.o_c + N * d * nx
        self.m = N * (d * nx + nx)
        lb = np.concatenate([np.tile(pb.x_lb, N), np.tile(pb.u_lb, N), np.tile(pb.x_lb, N * d)])
        ub = np.concatenate([np.tile(pb.x_ub, N), np.tile(pb.u_ub, N), np.tile(pb.x_ub, N * d)])
        r = o.bound_relax_factor
        self.lb = np.where(np.isfinite(lb), lb - r * np.maximum(1, np.abs(lb)), lb)
        self.ub = np.where(np.isfinite(ub), ub + r * np.maximum(1, np.abs(ub)), ub)
        self.has_l, self.has_u = np.isfinite(self.lb), np.isfinite(self.ub)

    def xcol(self, k):        # columns of x_k (k >= 1)
        return [(k - 1) * self.pb.nx + i for i in range(self.pb.nx)]

    def ucol(self, k):
        return [self.o_u + k * self.pb.nu + i for i in range(self.pb.nu)]

    def ccol(self, k, i):     # columns of collocation state i (1..d) of interval k
        pb = self.pb
        return [self.o_c + (k * pb.d + i - 1) * pb.nx + a for a in range(pb.nx)]

    def _unpack(self, w, x0):
        pb = self.pb
        B, N, nx, nu, d = w.shape[0], pb.N, pb.nx, pb.nu, pb.d
        X = np.concatenate([x0[:, None, :], w[:, :N * nx].reshape(B, N, nx)], axis=1)
        U = w[:, self.o_u:self.o_c].reshape(B, N, nu)
        Xc = w[:, self.o_c:].reshape(B, N, d, nx)
        return X, U, Xc

    def _cost(self, X, U, u_old, Xc=None, p=None):
        pb = self.pb
        f = np.zeros(X.shape[0])
        for k in range(pb.N):
            if pb.objective == 'continuous':
                for i in range(1, pb.d + 1):
                    f += pb.dt * pb.B[i] * pb.lagrange(Xc[:, k, i - 1], U[:, k], p, k, u_old)
            else:
                f += pb.lagrange(X[:, k], U[:, k], p, k, u_old)
        return f + pb.mayer(X[:, pb.N], p)

    def eval_fc(self, w, data):
        pb = self.pb
        x0, p, u_old = data['x0'], data['p'], data.get('u_old')
        X, U, Xc = self._unpack(w, x0)
        B, N, nx, d = w.shape[0], pb.N, pb.nx, pb.d
        c = np.empty((B, N, d + 1, nx))
        for k in range(N):
            xf = pb.D[0] * X[:, k]
            for i in range(1, d + 1):
                xp = pb.C[0, i] * X[:, k]
                for j in range(d):
                    xp = xp + pb.C[j + 1, i] * Xc[:, k, j]
                c[:, k, i - 1] = pb.dt * pb.rhs(Xc[:, k, i - 1], U[:, k], p) - xp
                xf = xf + pb.D[i] * Xc[:, k, i - 1]
            c[:, k, d] = X[:, k + 1] - xf
        return self._cost(X, U, u_old, Xc, p), c.reshape(B, -1)

    def eval_all(self, w, lam, data):
        pb = self.pb
        N, nx, nu, nz, d = pb.N, pb.nx, pb.nu, pb.nz, pb.d
        x0, p, u_old = data['x0'], data['p'], data.get('u_old')
        X, U, Xc = self._unpack(w, x0)
        B = w.shape[0]
        bi = np.arange(B)
        g = np.zeros((B, self.nw))
        c = np.empty((B, N, d + 1, nx))
        J = np.zeros((B, self.m, self.nw))
        W = np.zeros((B, self.nw, self.nw))
        lam = lam.reshape(B, N, d + 1, nx)
        mk = (d + 1) * nx
        eye = np.eye(nx)
        for k in range(N):
            if pb.objective == 'continuous':   # quadrature of the Lagrange term at the collocation states (modeling.py:1195)
                for i in range(1, d + 1):
                    _, gz, Hz = pb.lagrange(Xc[:, k, i - 1], U[:, k], p, k, u_old, need=1)
                    cols = self.ccol(k, i) + self.ucol(k)
                    wq = pb.dt * pb.B[i]
                    g[:, cols] += wq * gz
                    W[np.ix_(bi, cols, cols)] += wq * Hz
            else:                              # cost at the shooting node
                _, gz, Hz = pb.lagrange(X[:, k], U[:, k], p, k, u_old, need=1)
                cols = (self.xcol(k) if k > 0 else []) + self.ucol(k)
                sel = (list(range(nx)) if k > 0 else []) + list(range(nx, nz))
                g[:, cols] += gz[:, sel]
                W[np.ix_(bi, cols, cols)] += Hz[np.ix_(bi, sel, sel)]
            xf = pb.D[0] * X[:, k]
            for i in range(1, d + 1):
                rows = [k * mk + (i - 1) * nx + a for a in range(nx)]
                f, fw, fww = pb.rhs(Xc[:, k, i - 1], U[:, k], p, need=2)
                xp = pb.C[0, i] * X[:, k]
                if k > 0:
                    J[:, rows, self.xcol(k)] += -pb.C[0, i]
                for j in range(d):
                    xp = xp + pb.C[j + 1, i] * Xc[:, k, j]
                    J[:, rows, self.ccol(k, j + 1)] += -pb.C[j + 1, i]
                c[:, k, i - 1] = pb.dt * f - xp
                ci, uc = self.ccol(k, i), self.ucol(k)
                J[np.ix_(bi, rows, ci)] += pb.dt * fw[:, :, :nx]
                J[np.ix_(bi, rows, uc)] += pb.dt * fw[:, :, nx:]
                Hl = pb.dt * np.einsum('bm,bmzy->bzy', lam[:, k, i - 1], fww)
                both = ci + uc
                W[np.ix_(bi, both, both)] += Hl
                xf = xf + pb.D[i] * Xc[:, k, i - 1]
                rc = [k * mk + d * nx + a for a in range(nx)]
                J[:, rc, ci] += -pb.D[i]
            rc = [k * mk + d * nx + a for a in range(nx)]
            c[:, k, d] = X[:, k + 1] - xf
            J[:, rc, self.xcol(k + 1)] = 1.0
            if k > 0:
                J[:, rc, self.xcol(k)] += -pb.D[0]
        _, gN, HN = pb.mayer(X[:, N], p, need=1)
        g[:, self.xcol(N)] += gN
        W[np.ix_(bi, self.xcol(N), self.xcol(N))] += HN
        return self._cost(X, U, u_old, Xc, p), g, c.reshape(B, -1), J, W

    def solve(self, x0, p, w0=None, u_old=None, verbose=False):
        pb = self.pb
        x0 = np.atleast_2d(np.asarray(x0, dtype=float)) / pb.sx
        B = x0.shape[0]
        p = np.broadcast_to(np.atleast_2d(np.asarray(p, dtype=float)), (B, pb.np_)) if pb.np_ else np.zeros((B, 0))
        data = {'x0': x0, 'p': p}
        if u_old is not None:
            data['u_old'] = np.broadcast_to(np.atleast_2d(np.asarray(u_old, dtype=float)), (B, pb.nu))
        if w0 is None:
            w0 = np.concatenate([np.tile(pb.x_guess, pb.N), np.tile(pb.u_guess, pb.N), np.tile(pb.x_guess, pb.N * pb.d)])
        res = self.solve_data(data, w0, verbose)
        X, U, Xc = self._unpack(res['w'], x0)
        res.update(X=X, U=U, Xc=Xc, u0=U[:, 0] * pb.su, x0=x0)
        return res

    def to_v(self, res):
        B = res['X'].shape[0]
        return np.concatenate([res['X'].reshape(B, -1), res['U'].reshape(B, -1), res['Xc'].reshape(B, -1)], axis=1)

    def w_from_v(self, v):
        v = np.atleast_2d(v)
        return v[:, self.pb.nx:]
