"""Oracle: moving-horizon estimation - transcription + the dense interior-point solver of oracle/nmpc.py.

TEST INFRASTRUCTURE ONLY - never imported by the product package.   PARITY UNPINNED (see oracle/nmpc.py): the only
reference test that reaches the MHE solver (tests/test_MHE.py:723-749) uses unseeded noise and asserts nothing.

Restates `MovingHorizonEstimator.setup` for a pre-discretised model with `integration_method='discrete'` and state
noise (hilo_mpc/modules/estimator/mhe.py:596-790) with the model parameters pinned (p_lb = p_ub):
  v = [p | x_0..x_N | w_0..w_{N-1}]                                  (mhe.py:614-655)
  g_k = x_{k+1} - (Phi_s(x_k, u_meas_k, p) + w_k) = 0                 (mhe.py:733-740: the *scaled* noise variable is added
                                                                      to the scaled state)
  J   = arrival(x_0) at k = 0, stage(w_k, x_k, y_k) for k >= 1        (mhe.py:742-748: no stage cost at k = 0, y_0 unused)
  arrival = (x_0 sx - x_arr)^T Wx (x_0 sx - x_arr)                   (util/modeling.py:747-777; costs act on UN-scaled
  stage   = (h(x_k sx) - y_k)^T Wy (.) + (w_k sw)^T Ww (w_k sw)       quantities: `_setup` substitutes x -> x*x_scale, :665-672)
  u_meas enters the scaled model un-divided, i.e. the model sees u_meas * su (mhe.py:352 vs :242)
`estimate` returns x_N * sx, the one-step-ahead state (mhe.py:381-384); the arrival guess of the next call is the
previous solution's x_2 ("smoothing", mhe.py:254-256).
"""
from __future__ import annotations

import numpy as np
import sympy as sp

from .models import _lam
from .nmpc import DenseIpm, IpmOptions, _wmat
from .shooting import ShootingMap

INF = np.inf


class MheProblem:
    def __init__(self, model, dt, N, order=4, n_sub=1, Wx=None, Wy=None, Ww=None,
                 x_lb=None, x_ub=None, w_lb=None, w_ub=None, x_scaling=None, w_scaling=None, u_scaling=None,
                 x_guess=None, w_guess=None):
        self.model, self.dt, self.N = model, float(dt), int(N)
        nx, nu, ny = model.nx, model.nu, model.ny
        self.nx, self.nu, self.ny, self.np_ = nx, nu, ny, model.np_
        self.sx = np.ones(nx) if x_scaling is None else np.asarray(x_scaling, dtype=float)
        self.sw = np.ones(nx) if w_scaling is None else np.asarray(w_scaling, dtype=float)
        self.su = np.ones(nu) if u_scaling is None else np.asarray(u_scaling, dtype=float)
        self.Wx = _wmat(0. if Wx is None else Wx, nx)
        self.Wy = _wmat(0. if Wy is None else Wy, ny)
        self.Ww = _wmat(0. if Ww is None else Ww, nx)
        self.x_lb = (np.full(nx, -INF) if x_lb is None else np.asarray(x_lb, dtype=float)) / self.sx
        self.x_ub = (np.full(nx, INF) if x_ub is None else np.asarray(x_ub, dtype=float)) / self.sx
        self.w_lb = (np.full(nx, -INF) if w_lb is None else np.asarray(w_lb, dtype=float)) / self.sw
        self.w_ub = (np.full(nx, INF) if w_ub is None else np.asarray(w_ub, dtype=float)) / self.sw
        self.x_guess = (np.zeros(nx) if x_guess is None else np.asarray(x_guess, dtype=float)) / self.sx
        self.w_guess = (np.zeros(nx) if w_guess is None else np.asarray(w_guess, dtype=float)) / self.sw
        self.smap = ShootingMap(model, order, n_sub)
        args = [model.x, model.u, model.p, [model.dt]]
        H = sp.Matrix(model.meas)
        self._h = _lam(list(H), args)
        self._hx = _lam(H.jacobian(model.x).tolist(), args)
        self._hxx = _lam([[[sp.diff(H[m], a, b) for b in model.x] for a in model.x] for m in range(ny)], args)
        # bookkeeping of mhe.py:614-655 (bit-exact index maps)
        off = self.np_
        self.p_ind = [list(range(0, self.np_))] if self.np_ else []
        self.x_ind = []
        for _ in range(N + 1):
            self.x_ind.append(list(range(off, off + nx)))
            off += nx
        self.w_ind = []
        for _ in range(N):
            self.w_ind.append(list(range(off, off + nx)))
            off += nx
        self.n_v = off                                               # mhe.py:596-598
        self.n_g = N * nx

    def phi(self, xs, u, p, need=0):
        x = xs * self.sx
        ue = u * self.su
        if need == 0:
            return self.smap.value(x, ue, p, self.dt) / self.sx
        f, J, H = self.smap(x, ue, p, self.dt)
        nx = self.nx
        J = J[:, :, :nx] * self.sx[None, None, :] / self.sx[None, :, None]
        H = H[:, :, :nx, :nx] * self.sx[None, None, :, None] * self.sx[None, None, None, :] / self.sx[None, :, None, None]
        return f / self.sx, J, H

    def meas(self, xs, u, p, need=0):
        x = xs * self.sx
        ue = u * self.su
        h = self._h(x, ue, p, self.dt)
        if need == 0:
            return h
        hx = self._hx(x, ue, p, self.dt) * self.sx[None, None, :]
        hxx = self._hxx(x, ue, p, self.dt) * self.sx[None, None, :, None] * self.sx[None, None, None, :]
        return h, hx, hxx


class MheIpm(DenseIpm):
    """Free variables w = [x_0..x_N | w_0..w_{N-1}] (the pinned parameters are data)."""

    def __init__(self, prob: MheProblem, options: IpmOptions | None = None):
        self.pb = prob
        self.o = options or IpmOptions()
        N, nx = prob.N, prob.nx
        self.nw = (N + 1) * nx + N * nx
        self.m = N * nx
        self.ixs = [list(range(k * nx, (k + 1) * nx)) for k in range(N + 1)]
        self.iws = [list(range((N + 1) * nx + k * nx, (N + 1) * nx + (k + 1) * nx)) for k in range(N)]
        lb = np.concatenate([np.tile(prob.x_lb, N + 1), np.tile(prob.w_lb, N)])
        ub = np.concatenate([np.tile(prob.x_ub, N + 1), np.tile(prob.w_ub, N)])
        r = self.o.bound_relax_factor
        self.lb = np.where(np.isfinite(lb), lb - r * np.maximum(1, np.abs(lb)), lb)
        self.ub = np.where(np.isfinite(ub), ub + r * np.maximum(1, np.abs(ub)), ub)
        self.has_l, self.has_u = np.isfinite(self.lb), np.isfinite(self.ub)

    def _XW(self, w):
        pb = self.pb
        B = w.shape[0]
        X = w[:, :(pb.N + 1) * pb.nx].reshape(B, pb.N + 1, pb.nx)
        Wn = w[:, (pb.N + 1) * pb.nx:].reshape(B, pb.N, pb.nx)
        return X, Wn

    def eval_fc(self, w, data):
        pb = self.pb
        X, Wn = self._XW(w)
        B = w.shape[0]
        p, xa, um, ym = data['p'], data['x_arrival'], data['u_meas'], data['y_meas']
        d = X[:, 0] * pb.sx - xa
        f = np.einsum('bi,ij,bj->b', d, pb.Wx, d)
        c = np.empty((B, pb.N, pb.nx))
        for k in range(pb.N):
            c[:, k] = X[:, k + 1] - (pb.phi(X[:, k], um[:, k], p) + Wn[:, k])
            if k >= 1:
                r = pb.meas(X[:, k], um[:, k], p) - ym[:, k]
                ws = Wn[:, k] * pb.sw
                f += np.einsum('bi,ij,bj->b', r, pb.Wy, r) + np.einsum('bi,ij,bj->b', ws, pb.Ww, ws)
        return f, c.reshape(B, -1)

    def eval_all(self, w, lam, data):
        pb = self.pb
        N, nx = pb.N, pb.nx
        X, Wn = self._XW(w)
        B = w.shape[0]
        p, xa, um, ym = data['p'], data['x_arrival'], data['u_meas'], data['y_meas']
        g = np.zeros((B, self.nw))
        c = np.empty((B, N, nx))
        J = np.zeros((B, self.m, self.nw))
        W = np.zeros((B, self.nw, self.nw))
        lam = lam.reshape(B, N, nx)
        bi = np.arange(B)
        d = X[:, 0] * pb.sx - xa
        f = np.einsum('bi,ij,bj->b', d, pb.Wx, d)
        g[:, self.ixs[0]] += 2 * (d @ pb.Wx) * pb.sx
        W[np.ix_(bi, self.ixs[0], self.ixs[0])] += 2 * pb.Wx * np.outer(pb.sx, pb.sx)
        for k in range(N):
            Phi, Jk, Hk = pb.phi(X[:, k], um[:, k], p, need=2)
            c[:, k] = X[:, k + 1] - (Phi + Wn[:, k])
            rows = list(range(k * nx, (k + 1) * nx))
            J[np.ix_(bi, rows, self.ixs[k])] = -Jk
            J[:, rows, self.ixs[k + 1]] = 1.0
            J[:, rows, self.iws[k]] = -1.0
            W[np.ix_(bi, self.ixs[k], self.ixs[k])] -= np.einsum('bm,bmzy->bzy', lam[:, k], Hk)
            if k >= 1:
                h, hx, hxx = pb.meas(X[:, k], um[:, k], p, need=2)
                r = h - ym[:, k]
                rW = r @ pb.Wy
                f += np.einsum('bi,bi->b', rW, r)
                g[:, self.ixs[k]] += 2 * np.einsum('bm,bmz->bz', rW, hx)
                W[np.ix_(bi, self.ixs[k], self.ixs[k])] += 2 * np.einsum('bmz,mn,bny->bzy', hx, pb.Wy, hx) + \
                    2 * np.einsum('bm,bmzy->bzy', rW, hxx)
                ws = Wn[:, k] * pb.sw
                f += np.einsum('bi,ij,bj->b', ws, pb.Ww, ws)
                g[:, self.iws[k]] += 2 * (ws @ pb.Ww) * pb.sw
                W[np.ix_(bi, self.iws[k], self.iws[k])] += 2 * pb.Ww * np.outer(pb.sw, pb.sw)
        return f, g, c.reshape(B, -1), J, W

    def solve(self, x_arrival, p, u_meas, y_meas, w0=None, verbose=False):
        """x_arrival [B,nx]; p [B,np]; u_meas [B,N,nu]; y_meas [B,N,ny] (mhe.py:677-685 transposed to batch-major).
        Returns dict(..., X, Wn, x_opt = x_N * sx, v = reference layout [p | x | w])."""
        pb = self.pb
        xa = np.atleast_2d(np.asarray(x_arrival, dtype=float))
        B = xa.shape[0]
        p = np.broadcast_to(np.atleast_2d(np.asarray(p, dtype=float)), (B, pb.np_)) if pb.np_ else np.zeros((B, 0))
        um = np.asarray(u_meas, dtype=float).reshape(B, pb.N, pb.nu)
        ym = np.asarray(y_meas, dtype=float).reshape(B, pb.N, pb.ny)
        if w0 is None:
            w0 = np.concatenate([np.tile(pb.x_guess, pb.N + 1), np.tile(pb.w_guess, pb.N)])
        res = self.solve_data({'p': p, 'x_arrival': xa, 'u_meas': um, 'y_meas': ym}, w0, verbose)
        X, Wn = self._XW(res['w'])
        res.update(X=X, Wn=Wn, x_opt=X[:, -1] * pb.sx, v=np.concatenate([p, res['w']], axis=1))
        return res


# ======================================================================================================================
# parameter estimation: the model parameters are decision variables (mhe.py:614-623), bounded by p_lb / p_ub, with the
# arrival term (p - p_arrival)^T Wp (p - p_arrival) next to the state one (util/modeling.py:747-777, mhe.py:742-745).
# ======================================================================================================================
class MheEstProblem(MheProblem):
    """`est` = indices of the parameters that are estimated (p_lb < p_ub); the others are pinned through their bounds and
    enter as data, like IPOPT's treatment of fixed variables.  p_scaling as in mhe.py (the model sees p * sp)."""

    def __init__(self, model, dt, N, est, Wp=None, p_lb=None, p_ub=None, p_scaling=None, p_guess=None, **kw):
        from .models import OracleModel
        super().__init__(model, dt, N, **kw)
        self.est = list(est)
        ne = len(self.est)
        self.ne = ne
        self.sp = np.ones(ne) if p_scaling is None else np.asarray(p_scaling, dtype=float)
        self.Wp = _wmat(0. if Wp is None else Wp, ne)
        self.p_lb = (np.full(ne, -INF) if p_lb is None else np.asarray(p_lb, dtype=float)) / self.sp
        self.p_ub = (np.full(ne, INF) if p_ub is None else np.asarray(p_ub, dtype=float)) / self.sp
        self.p_guess = (np.zeros(ne) if p_guess is None else np.asarray(p_guess, dtype=float)) / self.sp
        # derivatives w.r.t. the estimated parameters: augment the model with them as constant states
        pe = [model.p[i] for i in self.est]
        aug = OracleModel(model.name + '_pest', model.model_id, model.x + pe, model.u,
                          [q for i, q in enumerate(model.p) if i not in self.est],
                          list(model.ode) + [sp.Integer(0)] * ne, model.meas, discrete=model.discrete, dt=model.dt)
        self.smap_aug = ShootingMap(aug, kw.get('order', 4), kw.get('n_sub', 1))
        args = [aug.x, aug.u, aug.p, [aug.dt]]
        H = sp.Matrix(aug.meas)
        self._ha = _lam(list(H), args)
        self._hax = _lam(H.jacobian(aug.x).tolist(), args)
        self._haxx = _lam([[[sp.diff(H[m], a, b) for b in aug.x] for a in aug.x] for m in range(model.ny)], args)
        self.fixed = [i for i in range(model.np_) if i not in self.est]

    def phi_aug(self, xs, pes, u, pfix, need=0):
        """scaled states xs, scaled estimated parameters pes -> Phi_s and derivatives w.r.t. (xs, pes)."""
        nx, ne = self.nx, self.ne
        xa = np.concatenate([xs * self.sx, pes * self.sp], axis=1)
        ue = u * self.su
        if need == 0:
            return self.smap_aug.value(xa, ue, pfix, self.dt)[:, :nx] / self.sx
        f, J, H = self.smap_aug(xa, ue, pfix, self.dt)
        sa = np.concatenate([self.sx, self.sp])
        na = nx + ne
        J = J[:, :nx, :na] * sa[None, None, :] / self.sx[None, :, None]
        H = H[:, :nx, :na, :na] * sa[None, None, :, None] * sa[None, None, None, :] / self.sx[None, :, None, None]
        return f[:, :nx] / self.sx, J, H

    def meas_aug(self, xs, pes, u, pfix, need=0):
        xa = np.concatenate([xs * self.sx, pes * self.sp], axis=1)
        ue = u * self.su
        h = self._ha(xa, ue, pfix, self.dt)
        if need == 0:
            return h
        sa = np.concatenate([self.sx, self.sp])
        return h, self._hax(xa, ue, pfix, self.dt) * sa[None, None, :], \
            self._haxx(xa, ue, pfix, self.dt) * sa[None, None, :, None] * sa[None, None, None, :]


class MheEstIpm(DenseIpm):
    """Free variables w = [p_est | x_0..x_N | w_0..w_{N-1}]."""

    def __init__(self, prob: MheEstProblem, options: IpmOptions | None = None):
        self.pb = pb = prob
        self.o = options or IpmOptions()
        N, nx, ne = pb.N, pb.nx, pb.ne
        self.o_x = ne
        self.o_w = ne + (N + 1) * nx
        self.nw = self.o_w + N * nx
        self.m = N * nx
        lb = np.concatenate([pb.p_lb, np.tile(pb.x_lb, N + 1), np.tile(pb.w_lb, N)])
        ub = np.concatenate([pb.p_ub, np.tile(pb.x_ub, N + 1), np.tile(pb.w_ub, N)])
        r = self.o.bound_relax_factor
        self.lb = np.where(np.isfinite(lb), lb - r * np.maximum(1, np.abs(lb)), lb)
        self.ub = np.where(np.isfinite(ub), ub + r * np.maximum(1, np.abs(ub)), ub)
        self.has_l, self.has_u = np.isfinite(self.lb), np.isfinite(self.ub)

    def _split(self, w):
        pb = self.pb
        B = w.shape[0]
        return w[:, :pb.ne], w[:, self.o_x:self.o_w].reshape(B, pb.N + 1, pb.nx), w[:, self.o_w:].reshape(B, pb.N, pb.nx)

    def eval_fc(self, w, data):
        pb = self.pb
        P, X, Wn = self._split(w)
        B = w.shape[0]
        pf, xa, pa, um, ym = data['p_fixed'], data['x_arrival'], data['p_arrival'], data['u_meas'], data['y_meas']
        d = X[:, 0] * pb.sx - xa
        dp = P * pb.sp - pa
        f = np.einsum('bi,ij,bj->b', d, pb.Wx, d) + np.einsum('bi,ij,bj->b', dp, pb.Wp, dp)
        c = np.empty((B, pb.N, pb.nx))
        for k in range(pb.N):
            c[:, k] = X[:, k + 1] - (pb.phi_aug(X[:, k], P, um[:, k], pf) + Wn[:, k])
            if k >= 1:
                r = pb.meas_aug(X[:, k], P, um[:, k], pf) - ym[:, k]
                ws = Wn[:, k] * pb.sw
                f += np.einsum('bi,ij,bj->b', r, pb.Wy, r) + np.einsum('bi,ij,bj->b', ws, pb.Ww, ws)
        return f, c.reshape(B, -1)

    def eval_all(self, w, lam, data):
        pb = self.pb
        N, nx, ne = pb.N, pb.nx, pb.ne
        P, X, Wn = self._split(w)
        B = w.shape[0]
        pf, xa, pa, um, ym = data['p_fixed'], data['x_arrival'], data['p_arrival'], data['u_meas'], data['y_meas']
        g = np.zeros((B, self.nw))
        c = np.empty((B, N, nx))
        J = np.zeros((B, self.m, self.nw))
        W = np.zeros((B, self.nw, self.nw))
        lam = lam.reshape(B, N, nx)
        bi = np.arange(B)
        pc = list(range(ne))
        xc = lambda k: [self.o_x + k * nx + i for i in range(nx)]      # noqa: E731
        wc = lambda k: [self.o_w + k * nx + i for i in range(nx)]      # noqa: E731
        d = X[:, 0] * pb.sx - xa
        dp = P * pb.sp - pa
        f = np.einsum('bi,ij,bj->b', d, pb.Wx, d) + np.einsum('bi,ij,bj->b', dp, pb.Wp, dp)
        g[:, xc(0)] += 2 * (d @ pb.Wx) * pb.sx
        W[np.ix_(bi, xc(0), xc(0))] += 2 * pb.Wx * np.outer(pb.sx, pb.sx)
        g[:, pc] += 2 * (dp @ pb.Wp) * pb.sp
        W[np.ix_(bi, pc, pc)] += 2 * pb.Wp * np.outer(pb.sp, pb.sp)
        for k in range(N):
            Phi, Jk, Hk = pb.phi_aug(X[:, k], P, um[:, k], pf, need=2)
            c[:, k] = X[:, k + 1] - (Phi + Wn[:, k])
            rows = list(range(k * nx, (k + 1) * nx))
            cols = xc(k) + pc                                         # order of the augmented (x, p_est)
            J[np.ix_(bi, rows, cols)] += -Jk
            J[:, rows, xc(k + 1)] += 1.0
            J[:, rows, wc(k)] += -1.0
            W[np.ix_(bi, cols, cols)] -= np.einsum('bm,bmzy->bzy', lam[:, k], Hk)
            if k >= 1:
                h, hx, hxx = pb.meas_aug(X[:, k], P, um[:, k], pf, need=2)
                r = h - ym[:, k]
                rW = r @ pb.Wy
                f += np.einsum('bi,bi->b', rW, r)
                g[:, cols] += 2 * np.einsum('bm,bmz->bz', rW, hx)
                W[np.ix_(bi, cols, cols)] += 2 * np.einsum('bmz,mn,bny->bzy', hx, pb.Wy, hx) + \
                    2 * np.einsum('bm,bmzy->bzy', rW, hxx)
                ws = Wn[:, k] * pb.sw
                f += np.einsum('bi,ij,bj->b', ws, pb.Ww, ws)
                g[:, wc(k)] += 2 * (ws @ pb.Ww) * pb.sw
                W[np.ix_(bi, wc(k), wc(k))] += 2 * pb.Ww * np.outer(pb.sw, pb.sw)
        return f, g, c.reshape(B, -1), J, W

    def solve(self, x_arrival, p_arrival, p_fixed, u_meas, y_meas, w0=None, verbose=False):
        """p_arrival [B,ne] (original units), p_fixed [B,np-ne]; returns dict(..., P (scaled), p_opt = P * sp, X, Wn,
        x_opt = x_N * sx, v = reference layout [p (all, scaled) | x | w])."""
        pb = self.pb
        xa = np.atleast_2d(np.asarray(x_arrival, dtype=float))
        B = xa.shape[0]
        pa = np.broadcast_to(np.atleast_2d(np.asarray(p_arrival, dtype=float)), (B, pb.ne))
        pf = np.broadcast_to(np.atleast_2d(np.asarray(p_fixed, dtype=float)), (B, len(pb.fixed))) if pb.fixed \
            else np.zeros((B, 0))
        um = np.asarray(u_meas, dtype=float).reshape(B, pb.N, pb.nu)
        ym = np.asarray(y_meas, dtype=float).reshape(B, pb.N, pb.ny)
        if w0 is None:
            w0 = np.concatenate([pb.p_guess, np.tile(pb.x_guess, pb.N + 1), np.tile(pb.w_guess, pb.N)])
        res = self.solve_data({'p_fixed': pf, 'x_arrival': xa, 'p_arrival': pa, 'u_meas': um, 'y_meas': ym}, w0, verbose)
        P, X, Wn = self._split(res['w'])
        pall = np.zeros((B, pb.np_))
        pall[:, pb.est] = P
        pall[:, pb.fixed] = pf
        res.update(P=P, p_opt=P * pb.sp, X=X, Wn=Wn, x_opt=X[:, -1] * pb.sx,
                   v=np.concatenate([pall, X.reshape(B, -1), Wn.reshape(B, -1)], axis=1))
        return res
