"""Oracle: NMPC of a semi-explicit DAE  dx/dt = f(x, z, u, p),  0 = g(x, z, u, p)  with the reference's default transcription,
direct collocation, in its SIMULTANEOUS form - collocation states AND the algebraic states at the collocation points are
decision variables, exactly like the reference.

TEST INFRASTRUCTURE ONLY - never imported by the product package.   PARITY UNPINNED for the DAE extension (the reference's DAE
test, tests/test_NMPC.py:1866-1987, asserts no number); the interior-point method and the ODE collocation it extends are
pinned by the CSTR notebook (oracle/nmpc_coll.py).

Restated from hilo_mpc/util/modeling.py:1128-1211 (`_collocation`: per collocation point i = 1..d the rows
[dt f(x_i, z_i, u) - sum_j C[j, i] x_j  |  g(x_i, z_i, u)], in that order, :1183-1190) and hilo_mpc/modules/controller/mpc.py:
  v = [x_0..x_N | u_0..u_{N-1} | z_0..z_N | (ip_k, zp_k) per interval]           (:1462-1518; n_zik = degree * n_z, :1322)
  g per interval = [collocation rows (above) | continuity x_{k+1} - sum_j D_j x_j]  (:1657-1669)
  the node blocks z_0..z_N enter no constraint and no cost: they stay where the initial guess puts them (bounds +-inf by
  default, mpc.py:645-701), so they are not carried as variables here and `to_v` fills the guess in.
Derivatives: one sympy function per interval (rows, Jacobian, multiplier-contracted Hessian w.r.t. the interval's variables).
"""
from __future__ import annotations

import numpy as np
import sympy as sp

from .models import _lam
from .nmpc import DenseIpm, IpmOptions
from .nmpc_coll import CollNmpcProblem, polynomial_basis


class DaeCollProblem(CollNmpcProblem):
    def __init__(self, model, dt, N, degree=3, points='radau', objective='continuous', z_guess=None, z_lb=None, z_ub=None,
                 **kw):
        assert model.z, "use CollNmpcProblem for an ODE"
        super().__init__(model, dt, N, degree=degree, points=points, objective=objective, **kw)
        nx, nu, nza, d = self.nx, self.nu, len(model.z), degree
        self.nza = nza
        self.z_guess = np.zeros(nza) if z_guess is None else np.asarray(z_guess, dtype=float)
        self.z_lb = np.full(nza, -np.inf) if z_lb is None else np.asarray(z_lb, dtype=float)
        self.z_ub = np.full(nza, np.inf) if z_ub is None else np.asarray(z_ub, dtype=float)
        # reference layout (mpc.py:1462-1518)
        off = (N + 1) * nx + N * nu
        self.z_ind = [list(range(off + k * nza, off + (k + 1) * nza)) for k in range(N + 1)]
        off += (N + 1) * nza
        self.ip_ind, self.zp_ind = [], []
        for k in range(N):
            self.ip_ind.append(list(range(off, off + d * nx)))
            off += d * nx
            self.zp_ind.append(list(range(off, off + d * nza)))
            off += d * nza
        self.n_v = off
        self.n_g = N * (d * (nx + nza) + nx)
        # ---- one interval symbolically: variables q = [x_k | u_k | Xc (d nx) | Zc (d nza)], rows R ----
        m = model
        xk = [sp.Symbol(f'xk{i}') for i in range(nx)]
        uk = [sp.Symbol(f'uk{i}') for i in range(nu)]
        Xc = [[sp.Symbol(f'xc{i}_{a}') for a in range(nx)] for i in range(d)]
        Zc = [[sp.Symbol(f'zc{i}_{a}') for a in range(nza)] for i in range(d)]
        lam = [sp.Symbol(f'l{r}') for r in range(d * (nx + nza))]
        pts = [xk] + Xc
        R = []
        for i in range(1, d + 1):
            sub = {**{m.x[a]: self.sx[a] * Xc[i - 1][a] for a in range(nx)}, **{m.u[a]: self.su[a] * uk[a] for a in range(nu)},
                   **{m.z[a]: Zc[i - 1][a] for a in range(nza)}}
            f = [e.subs(sub, simultaneous=True) / self.sx[a] for a, e in enumerate(m.ode)]       # base.py:1562-1591
            g = [e.subs(sub, simultaneous=True) for e in m.alg]
            for a in range(nx):
                R.append(self.dt * f[a] - sum(self.C[j, i] * pts[j][a] for j in range(d + 1)))
            R += g
        q = xk + uk + [s for row in Xc for s in row] + [s for row in Zc for s in row]
        args = [q, m.p, lam]
        self.nq = len(q)
        self._R = _lam(R, args)
        self._JR = _lam(sp.Matrix(R).jacobian(q).tolist(), args)
        L = sum(l * r for l, r in zip(lam, R))
        self._HR = _lam(sp.hessian(L, q).tolist(), args)


class DaeCollIpm(DenseIpm):
    """Free variables w = [x_1..x_N | u_0..u_{N-1} | (Xc_k, Zc_k) per interval]."""

    def __init__(self, prob: DaeCollProblem, options: IpmOptions | None = None):
        self.pb = pb = prob
        self.o = o = options or IpmOptions()
        N, nx, nu, d, nza = pb.N, pb.nx, pb.nu, pb.d, pb.nza
        self.o_u = N * nx
        self.o_c = self.o_u + N * nu
        self.blk = d * (nx + nza)
        self.nw = self.o_c + N * self.blk
        self.mk = self.blk + nx
        self.m = N * self.mk
        lb = np.concatenate([np.tile(pb.x_lb, N), np.tile(pb.u_lb, N)] + [np.concatenate([np.tile(pb.x_lb, d), np.tile(pb.z_lb, d)])] * N)
        ub = np.concatenate([np.tile(pb.x_ub, N), np.tile(pb.u_ub, N)] + [np.concatenate([np.tile(pb.x_ub, d), np.tile(pb.z_ub, d)])] * N)
        r = o.bound_relax_factor
        self.lb = np.where(np.isfinite(lb), lb - r * np.maximum(1, np.abs(lb)), lb)
        self.ub = np.where(np.isfinite(ub), ub + r * np.maximum(1, np.abs(ub)), ub)
        self.has_l, self.has_u = np.isfinite(self.lb), np.isfinite(self.ub)

    def xcol(self, k):
        return [(k - 1) * self.pb.nx + i for i in range(self.pb.nx)]

    def ucol(self, k):
        return [self.o_u + k * self.pb.nu + i for i in range(self.pb.nu)]

    def bcol(self, k):       # the interval's collocation block [Xc | Zc]
        return list(range(self.o_c + k * self.blk, self.o_c + (k + 1) * self.blk))

    def _unpack(self, w, x0):
        pb = self.pb
        B, N, nx, nu, d, nza = w.shape[0], pb.N, pb.nx, pb.nu, pb.d, pb.nza
        X = np.concatenate([x0[:, None, :], w[:, :N * nx].reshape(B, N, nx)], axis=1)
        U = w[:, self.o_u:self.o_c].reshape(B, N, nu)
        blk = w[:, self.o_c:].reshape(B, N, self.blk)
        return X, U, blk[:, :, :d * nx].reshape(B, N, d, nx), blk[:, :, d * nx:].reshape(B, N, d, nza)

    def _q(self, X, U, blk, k):
        return np.concatenate([X[:, k], U[:, k], blk[:, k]], axis=1)

    def _cost(self, X, U, Xc, p, u_old):
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
        X, U, Xc, Zc = self._unpack(w, x0)
        B, N, nx, d = w.shape[0], pb.N, pb.nx, pb.d
        blk = w[:, self.o_c:].reshape(B, N, self.blk)
        c = np.empty((B, N, self.mk))
        lam0 = np.zeros((B, self.blk))
        for k in range(N):
            c[:, k, :self.blk] = self._R(self._q(X, U, blk, k), p, lam0)
            xf = pb.D[0] * X[:, k]
            for i in range(1, d + 1):
                xf = xf + pb.D[i] * Xc[:, k, i - 1]
            c[:, k, self.blk:] = X[:, k + 1] - xf
        return self._cost(X, U, Xc, p, u_old), c.reshape(B, -1)

    _R = property(lambda s: s.pb._R)

    def eval_all(self, w, lam, data):
        pb = self.pb
        N, nx, nu, nz, d, nza = pb.N, pb.nx, pb.nu, pb.nz, pb.d, pb.nza
        x0, p, u_old = data['x0'], data['p'], data.get('u_old')
        X, U, Xc, Zc = self._unpack(w, x0)
        B = w.shape[0]
        bi = np.arange(B)
        blk = w[:, self.o_c:].reshape(B, N, self.blk)
        g = np.zeros((B, self.nw))
        c = np.empty((B, N, self.mk))
        J = np.zeros((B, self.m, self.nw))
        W = np.zeros((B, self.nw, self.nw))
        lam = lam.reshape(B, N, self.mk)
        for k in range(N):
            # cost: quadrature at the collocation states (continuous objective) or the node value
            if pb.objective == 'continuous':
                for i in range(1, d + 1):
                    _, gz, Hz = pb.lagrange(Xc[:, k, i - 1], U[:, k], p, k, u_old, need=1)
                    cols = self.bcol(k)[(i - 1) * nx:i * nx] + self.ucol(k)
                    wq = pb.dt * pb.B[i]
                    g[:, cols] += wq * gz
                    W[np.ix_(bi, cols, cols)] += wq * Hz
            else:
                _, gz, Hz = pb.lagrange(X[:, k], U[:, k], p, k, u_old, need=1)
                cols = (self.xcol(k) if k > 0 else []) + self.ucol(k)
                sel = (list(range(nx)) if k > 0 else []) + list(range(nx, nz))
                g[:, cols] += gz[:, sel]
                W[np.ix_(bi, cols, cols)] += Hz[np.ix_(bi, sel, sel)]
            q = self._q(X, U, blk, k)
            lk = lam[:, k, :self.blk]
            c[:, k, :self.blk] = pb._R(q, p, lk)
            JR, HR = pb._JR(q, p, lk), pb._HR(q, p, lk)
            qcols = (self.xcol(k) if k > 0 else [-1] * nx) + self.ucol(k) + self.bcol(k)
            keep = [j for j, cix in enumerate(qcols) if cix >= 0]
            cols = [qcols[j] for j in keep]
            rows = list(range(k * self.mk, k * self.mk + self.blk))
            J[np.ix_(bi, rows, cols)] += JR[:, :, keep]
            W[np.ix_(bi, cols, cols)] += HR[np.ix_(bi, keep, keep)]
            # continuity
            rc = list(range(k * self.mk + self.blk, (k + 1) * self.mk))
            xf = pb.D[0] * X[:, k]
            for i in range(1, d + 1):
                xf = xf + pb.D[i] * Xc[:, k, i - 1]
                J[:, rc, self.bcol(k)[(i - 1) * nx:i * nx]] += -pb.D[i]
            c[:, k, self.blk:] = X[:, k + 1] - xf
            J[:, rc, self.xcol(k + 1)] = 1.0
            if k > 0:
                J[:, rc, self.xcol(k)] += -pb.D[0]
        _, gN, HN = pb.mayer(X[:, N], p, need=1)
        g[:, self.xcol(N)] += gN
        W[np.ix_(bi, self.xcol(N), self.xcol(N))] += HN
        return self._cost(X, U, Xc, p, u_old), g, c.reshape(B, -1), J, W

    def solve(self, x0, p, w0=None, u_old=None, verbose=False):
        pb = self.pb
        x0 = np.atleast_2d(np.asarray(x0, dtype=float)) / pb.sx
        B = x0.shape[0]
        p = np.broadcast_to(np.atleast_2d(np.asarray(p, dtype=float)), (B, pb.np_)) if pb.np_ else np.zeros((B, 0))
        data = {'x0': x0, 'p': p}
        if u_old is not None:
            data['u_old'] = np.broadcast_to(np.atleast_2d(np.asarray(u_old, dtype=float)), (B, pb.nu))
        if w0 is None:
            w0 = np.concatenate([np.tile(pb.x_guess, pb.N), np.tile(pb.u_guess, pb.N)] +
                                [np.concatenate([np.tile(pb.x_guess, pb.d), np.tile(pb.z_guess, pb.d)])] * pb.N)
        res = self.solve_data(data, w0, verbose)
        X, U, Xc, Zc = self._unpack(res['w'], x0)
        res.update(X=X, U=U, Xc=Xc, Zc=Zc, u0=U[:, 0] * pb.su, x0=x0)
        return res

    def to_v(self, res):
        """Reference layout [x | u | z (node blocks: the guess) | (ip_k, zp_k)...]."""
        pb = self.pb
        B = res['X'].shape[0]
        parts = [res['X'].reshape(B, -1), res['U'].reshape(B, -1), np.tile(pb.z_guess, (B, pb.N + 1))]
        for k in range(pb.N):
            parts += [res['Xc'][:, k].reshape(B, -1), res['Zc'][:, k].reshape(B, -1)]
        return np.concatenate(parts, axis=1)
