"""Oracle: moving-horizon estimation with the reference's DEFAULT transcription - direct collocation.

TEST INFRASTRUCTURE ONLY - never imported by the product package.   PARITY UNPINNED (see oracle/mhe.py): the reference's tests
hold no number for the MHE; the collocation scheme itself is the pinned one of oracle/nmpc_coll.py (CSTR notebook), the
interior-point method the one of oracle/nmpc.py, and the solves are cross-checked by the KKT residual and against the
explicit-integrator transcription at a fine step (tests/test_oracle_mhe_coll.py).

Restates `MovingHorizonEstimator.setup` for a CONTINUOUS model with `integration_method='collocation'` (the default,
hilo_mpc/modules/estimator/mhe.py:512-593, optimizer options 'radau', degree 3) and state noise, parameters pinned:
  v = [p | x_0..x_N | w_0..w_{N-1} | ip_0..ip_{N-1}]           mhe.py:614-671 (collocation states AFTER the noise block)
  per interval k, collocation states X_{k,1..d} (bounds / guess = the state's, tiled, mhe.py:523-525) with
      dt f(X_{k,i}, u_meas_k, p) - sum_j C[j, i] X_{k,j} = 0,   X_{k,0} = x_k                       (modeling.py:1183-1189)
  and  x_{k+1} - (sum_j D_j X_{k,j} + w_k) = 0                  mhe.py:726-733: the noise is added to the integrated state
  g per stage = [collocation rows (d nx) | continuity (nx)]     mhe.py:728, :740
  J as in oracle/mhe.py (arrival at k = 0, measurement + noise terms for k >= 1 at the shooting nodes, mhe.py:742-748).
"""
from __future__ import annotations

import numpy as np

from .mhe import MheProblem
from .nmpc import DenseIpm, IpmOptions
from .nmpc_coll import polynomial_basis


class MheCollProblem(MheProblem):
    def __init__(self, model, dt, N, degree=3, points='radau', **kw):
        assert not model.discrete, "collocation needs the continuous model"
        kw.setdefault('order', 4)
        super().__init__(model, dt, N, **kw)
        self.d = degree
        self.B, self.C, self.D, self.tau = polynomial_basis(degree, points)
        nx, d = self.nx, degree
        off = self.n_v                                              # behind [p | x | w]
        self.ip_ind = [list(range(off + k * d * nx, off + (k + 1) * d * nx)) for k in range(N)]
        self.n_v = off + N * d * nx
        self.n_g = N * (d * nx + nx)

    def rhs(self, xs, u, p, need=0):
        """Scaled continuous right-hand side f(xs sx, u su, p) / sx with derivatives w.r.t. xs."""
        x = xs * self.sx
        ue = np.broadcast_to(u * self.su, (x.shape[0], self.nu))
        sm = self.smap
        if need == 0:
            return sm._f(x, ue, p, self.dt) / self.sx
        f, fw, fww = sm._rhs(x, ue, p, self.dt)
        nx = self.nx
        fx = fw[:, :, :nx] * self.sx[None, None, :] / self.sx[None, :, None]
        fxx = fww[:, :, :nx, :nx] * self.sx[None, None, :, None] * self.sx[None, None, None, :] / self.sx[None, :, None, None]
        return f / self.sx, fx, fxx


class MheCollIpm(DenseIpm):
    """Free variables w = [x_0..x_N | w_0..w_{N-1} | X_0..X_{N-1}] (the pinned parameters are data)."""

    def __init__(self, prob: MheCollProblem, options: IpmOptions | None = None):
        self.pb = pb = prob
        self.o = o = options or IpmOptions()
        N, nx, d = pb.N, pb.nx, pb.d
        self.o_w = (N + 1) * nx
        self.o_c = self.o_w + N * nx
        self.nw = self.o_c + N * d * nx
        self.m = N * (d * nx + nx)
        lb = np.concatenate([np.tile(pb.x_lb, N + 1), np.tile(pb.w_lb, N), np.tile(pb.x_lb, N * d)])
        ub = np.concatenate([np.tile(pb.x_ub, N + 1), np.tile(pb.w_ub, N), np.tile(pb.x_ub, N * d)])
        r = o.bound_relax_factor
        self.lb = np.where(np.isfinite(lb), lb - r * np.maximum(1, np.abs(lb)), lb)
        self.ub = np.where(np.isfinite(ub), ub + r * np.maximum(1, np.abs(ub)), ub)
        self.has_l, self.has_u = np.isfinite(self.lb), np.isfinite(self.ub)

    def xcol(self, k):
        return [k * self.pb.nx + i for i in range(self.pb.nx)]

    def wcol(self, k):
        return [self.o_w + k * self.pb.nx + i for i in range(self.pb.nx)]

    def ccol(self, k, i):     # collocation state i (1..d) of interval k
        pb = self.pb
        return [self.o_c + (k * pb.d + i - 1) * pb.nx + a for a in range(pb.nx)]

    def _unpack(self, w):
        pb = self.pb
        B, N, nx, d = w.shape[0], pb.N, pb.nx, pb.d
        return (w[:, :self.o_w].reshape(B, N + 1, nx), w[:, self.o_w:self.o_c].reshape(B, N, nx),
                w[:, self.o_c:].reshape(B, N, d, nx))

    def eval_fc(self, w, data):
        pb = self.pb
        X, Wn, Xc = self._unpack(w)
        B, N, nx, d = w.shape[0], pb.N, pb.nx, pb.d
        p, xa, um, ym = data['p'], data['x_arrival'], data['u_meas'], data['y_meas']
        dd = X[:, 0] * pb.sx - xa
        f = np.einsum('bi,ij,bj->b', dd, pb.Wx, dd)
        c = np.empty((B, N, d + 1, nx))
        for k in range(N):
            xf = pb.D[0] * X[:, k]
            for i in range(1, d + 1):
                xp = pb.C[0, i] * X[:, k]
                for j in range(d):
                    xp = xp + pb.C[j + 1, i] * Xc[:, k, j]
                c[:, k, i - 1] = pb.dt * pb.rhs(Xc[:, k, i - 1], um[:, k], p) - xp
                xf = xf + pb.D[i] * Xc[:, k, i - 1]
            c[:, k, d] = X[:, k + 1] - (xf + Wn[:, k])
            if k >= 1:
                r = pb.meas(X[:, k], um[:, k], p) - ym[:, k]
                ws = Wn[:, k] * pb.sw
                f += np.einsum('bi,ij,bj->b', r, pb.Wy, r) + np.einsum('bi,ij,bj->b', ws, pb.Ww, ws)
        return f, c.reshape(B, -1)

    def eval_all(self, w, lam, data):
        pb = self.pb
        N, nx, d = pb.N, pb.nx, pb.d
        X, Wn, Xc = self._unpack(w)
        B = w.shape[0]
        bi = np.arange(B)
        p, xa, um, ym = data['p'], data['x_arrival'], data['u_meas'], data['y_meas']
        g = np.zeros((B, self.nw))
        c = np.empty((B, N, d + 1, nx))
        J = np.zeros((B, self.m, self.nw))
        W = np.zeros((B, self.nw, self.nw))
        lam = lam.reshape(B, N, d + 1, nx)
        mk = (d + 1) * nx
        dd = X[:, 0] * pb.sx - xa
        f = np.einsum('bi,ij,bj->b', dd, pb.Wx, dd)
        g[:, self.xcol(0)] += 2 * (dd @ pb.Wx) * pb.sx
        W[np.ix_(bi, self.xcol(0), self.xcol(0))] += 2 * pb.Wx * np.outer(pb.sx, pb.sx)
        for k in range(N):
            xf = pb.D[0] * X[:, k]
            rc = [k * mk + d * nx + a for a in range(nx)]
            for i in range(1, d + 1):
                rows = [k * mk + (i - 1) * nx + a for a in range(nx)]
                fv, fx, fxx = pb.rhs(Xc[:, k, i - 1], um[:, k], p, need=2)
                xp = pb.C[0, i] * X[:, k]
                J[:, rows, self.xcol(k)] += -pb.C[0, i]
                for j in range(d):
                    xp = xp + pb.C[j + 1, i] * Xc[:, k, j]
                    J[:, rows, self.ccol(k, j + 1)] += -pb.C[j + 1, i]
                c[:, k, i - 1] = pb.dt * fv - xp
                ci = self.ccol(k, i)
                J[np.ix_(bi, rows, ci)] += pb.dt * fx
                W[np.ix_(bi, ci, ci)] += pb.dt * np.einsum('bm,bmzy->bzy', lam[:, k, i - 1], fxx)
                xf = xf + pb.D[i] * Xc[:, k, i - 1]
                J[:, rc, ci] += -pb.D[i]
            c[:, k, d] = X[:, k + 1] - (xf + Wn[:, k])
            J[:, rc, self.xcol(k + 1)] += 1.0
            J[:, rc, self.xcol(k)] += -pb.D[0]
            J[:, rc, self.wcol(k)] += -1.0
            if k >= 1:
                h, hx, hxx = pb.meas(X[:, k], um[:, k], p, need=2)
                r = h - ym[:, k]
                rW = r @ pb.Wy
                f += np.einsum('bi,bi->b', rW, r)
                g[:, self.xcol(k)] += 2 * np.einsum('bm,bmz->bz', rW, hx)
                W[np.ix_(bi, self.xcol(k), self.xcol(k))] += 2 * np.einsum('bmz,mn,bny->bzy', hx, pb.Wy, hx) + \
                    2 * np.einsum('bm,bmzy->bzy', rW, hxx)
                ws = Wn[:, k] * pb.sw
                f += np.einsum('bi,ij,bj->b', ws, pb.Ww, ws)
                g[:, self.wcol(k)] += 2 * (ws @ pb.Ww) * pb.sw
                W[np.ix_(bi, self.wcol(k), self.wcol(k))] += 2 * pb.Ww * np.outer(pb.sw, pb.sw)
        return f, g, c.reshape(B, -1), J, W

    def solve(self, x_arrival, p, u_meas, y_meas, w0=None, verbose=False):
        """As MheIpm.solve; `v` = reference layout [p | x | w | ip], `lam` in the reference's g order."""
        pb = self.pb
        xa = np.atleast_2d(np.asarray(x_arrival, dtype=float))
        B = xa.shape[0]
        p = np.broadcast_to(np.atleast_2d(np.asarray(p, dtype=float)), (B, pb.np_)) if pb.np_ else np.zeros((B, 0))
        um = np.asarray(u_meas, dtype=float).reshape(B, pb.N, pb.nu)
        ym = np.asarray(y_meas, dtype=float).reshape(B, pb.N, pb.ny)
        if w0 is None:
            w0 = np.concatenate([np.tile(pb.x_guess, pb.N + 1), np.tile(pb.w_guess, pb.N), np.tile(pb.x_guess, pb.N * pb.d)])
        res = self.solve_data({'p': p, 'x_arrival': xa, 'u_meas': um, 'y_meas': ym}, w0, verbose)
        X, Wn, Xc = self._unpack(res['w'])
        res.update(X=X, Wn=Wn, Xc=Xc, x_opt=X[:, -1] * pb.sx, v=np.concatenate([p, res['w']], axis=1))
        return res
