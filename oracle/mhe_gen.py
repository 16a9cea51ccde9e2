"""Oracle: moving-horizon estimation in the variants the reference's own tests configure (tests/test_MHE.py:20-110, :150-230, :331)
- WITHOUT state noise, with ESTIMATED parameters, under collocation (the default for a continuous model) or with a discrete model -
in one general transcription on the dense interior-point solver of oracle/nmpc.py.

TEST INFRASTRUCTURE ONLY - never imported by the product package.   PARITY UNPINNED: the reference's MHE tests assert no number.
tests/test_oracle_mhe_gen.py checks this module against oracle/mhe.py (MheIpm, MheEstIpm) and oracle/mhe_coll.py on the cases
those cover, its derivatives against finite differences, and scipy SLSQP on the same NLP.

Restated from hilo_mpc/modules/estimator/mhe.py:596-790 (`_setup`):
  v = [p | x_0..x_N | w_0..w_{N-1} (only with state noise, :599, :636-645) | ip_0..ip_{N-1} (collocation, :647-660)]
  per interval: [collocation equations dt f(x_{k,i}, u_k, p) - sum_j C[j,i] x_{k,j} (modeling.py:1183-1189) | x_{k+1} - x_end (- w_k)]
        x_end = sum_j D_j x_{k,j} (collocation, :726-731) or the discrete map (:733-736); the noise is added to it (:731, :736)
  J = arrival(x_0, p) at k = 0 (:742-745; modeling.py:747-777: states and ALL parameters), for k >= 1 the stage term
      (h(x_k) - y_k)' Wy (.) [+ w_k' Ww w_k] (:746-748) - no stage term at k = 0, none at x_N
  the parameters are ONE vector of variables with bounds p_lb / p_ub (:614-623); p_lb = p_ub pins one (IPOPT removes it)
  costs act on un-scaled quantities, the noise is added to the scaled state (oracle/mhe.py restates both).
  stage constraint (`mhe.stage_constraint`, :498-508): HARD rows lb <= c(x, p) <= ub at every collocation point (:536-553, in front of
      the collocation equations) and at every node k < N (:749-757, behind the continuity rows).  QUIRKS restated: the estimator never
      calls the constraint's `_check_and_setup` - the expression is evaluated on the SCALED variables - and the soft branch cannot run
      (its penalty function is never created).  Rows the IPOPT way: a slack per row with (relaxed) bounds.
Without state noise the trajectory is a function of (x_0, p) alone - mhe.py:717-728 raises for 'multiple_shooting' only; the
collocation and discrete branches run (what tests/test_MHE.py:20-110 configure).
"""
from __future__ import annotations

import numpy as np
import sympy as sp

from .models import _lam
from .nmpc import DenseIpm, IpmOptions, _wmat
from .nmpc_coll import polynomial_basis

INF = np.inf


class MheGenProblem:
    """degree = 0: the model's discrete map (a discrete model, or `order` = explicit Runge-Kutta order of a continuous one,
    SURVEY Q19); degree >= 1: collocation.  est: indices of the estimated parameters; noise: state noise variables w_k."""

    def __init__(self, model, dt, N, degree=3, points='radau', order=4, noise=True, est=(), Wx=None, Wp=None, Wy=None, Ww=None,
                 x_lb=None, x_ub=None, w_lb=None, w_ub=None, p_lb=None, p_ub=None, x_scaling=None, w_scaling=None, u_scaling=None,
                 p_scaling=None, x_guess=None, w_guess=None, p_guess=None, constraint=None):
        self.model, self.dt, self.N, self.d, self.noise = model, float(dt), int(N), int(degree), bool(noise)
        m = model
        nx, nu, ny, npar = m.nx, m.nu, m.ny, m.np_
        self.nx, self.nu, self.ny, self.np_ = nx, nu, ny, npar
        self.est = list(est)
        self.fixed = [i for i in range(npar) if i not in self.est]
        ne = self.ne = len(self.est)
        self.sx = np.ones(nx) if x_scaling is None else np.asarray(x_scaling, dtype=float)
        self.sw = np.ones(nx) if w_scaling is None else np.asarray(w_scaling, dtype=float)
        self.su = np.ones(nu) if u_scaling is None else np.asarray(u_scaling, dtype=float)
        self.sp = np.ones(ne) if p_scaling is None else np.asarray(p_scaling, dtype=float)
        self.Wx, self.Wy = _wmat(0. if Wx is None else Wx, nx), _wmat(0. if Wy is None else Wy, ny)
        self.Ww = _wmat(0. if Ww is None else Ww, nx)
        self.Wp = _wmat(0. if Wp is None else Wp, ne) if ne else np.zeros((0, 0))
        box = lambda v, n, s, dflt: (np.full(n, dflt) if v is None else np.asarray(v, dtype=float)) / s      # noqa: E731
        self.x_lb, self.x_ub = box(x_lb, nx, self.sx, -INF), box(x_ub, nx, self.sx, INF)
        self.w_lb, self.w_ub = box(w_lb, nx, self.sw, -INF), box(w_ub, nx, self.sw, INF)
        self.p_lb, self.p_ub = box(p_lb, ne, self.sp, -INF), box(p_ub, ne, self.sp, INF)
        self.x_guess, self.w_guess, self.p_guess = box(x_guess, nx, self.sx, 0.), box(w_guess, nx, self.sw, 0.), box(p_guess, ne, self.sp, 0.)
        d = self.d
        if d:
            assert not m.discrete
            self.B, self.C, self.D, self.tau = polynomial_basis(d, points)
            rhs = list(m.ode)
        else:
            md = m if m.discrete else m.discretize(order)
            rhs = [e.subs(md.dt, self.dt) for e in md.ode]
        # ---- one interval symbolically (scaled variables) ----
        xk = [sp.Symbol(f'xk{i}') for i in range(nx)]
        xn = [sp.Symbol(f'xn{i}') for i in range(nx)]
        wk = [sp.Symbol(f'wk{i}') for i in range(nx)] if self.noise else []
        pe = [sp.Symbol(f'pe{i}') for i in range(ne)]
        Xc = [[sp.Symbol(f'xc{i}_{a}') for a in range(nx)] for i in range(d)]
        um = [sp.Symbol(f'um{i}') for i in range(nu)]
        ym = [sp.Symbol(f'ym{i}') for i in range(ny)]
        pf = [sp.Symbol(f'pf{i}') for i in range(len(self.fixed))]
        xa = [sp.Symbol(f'xa{i}') for i in range(nx)]
        pa = [sp.Symbol(f'pa{i}') for i in range(ne)]
        first, inner = sp.Symbol('first_interval'), sp.Symbol('inner_interval')      # k = 0 / k >= 1

        def at(xs):
            sub = {m.x[a]: self.sx[a] * xs[a] for a in range(nx)}
            sub.update({m.u[a]: self.su[a] * um[a] for a in range(nu)})                  # mhe.py:352 vs :242
            for j, i in enumerate(self.est):
                sub[m.p[i]] = self.sp[j] * pe[j]
            for j, i in enumerate(self.fixed):
                sub[m.p[i]] = pf[j]
            return sub
        # constraint = dict(expr=[sympy expressions / strings of the model's state and parameter symbols], lb=[...], ub=[...])
        self.rows = []                                               # (expression index, lb, ub)
        cexpr = []
        if constraint:
            names = {str(q): q for q in m.x + m.p}
            cexpr = [sp.sympify(e, locals=names) if isinstance(e, str) else sp.sympify(e) for e in constraint['expr']]
            lbc = np.broadcast_to(np.asarray(constraint.get('lb', -INF), dtype=float), (len(cexpr),))
            ubc = np.broadcast_to(np.asarray(constraint.get('ub', INF), dtype=float), (len(cexpr),))
            self.rows = [(j, lbc[j], ubc[j]) for j in range(len(cexpr)) if np.isfinite(lbc[j]) or np.isfinite(ubc[j])]
        self.ncon, self.nrow = len(cexpr), len(self.rows)
        ss = [sp.Symbol(f's{r}') for r in range((d + 1) * self.nrow)]

        def con_rows(xs, sl):
            sub = {m.x[a]: xs[a] for a in range(nx)}                                     # SCALED variables (no _check_and_setup)
            for j, i in enumerate(self.est):
                sub[m.p[i]] = pe[j]
            for j, i in enumerate(self.fixed):
                sub[m.p[i]] = pf[j]
            return [cexpr[j].subs(sub, simultaneous=True) - sl[r] for r, (j, _, _) in enumerate(self.rows)]
        R = []
        for i in range(d):
            R += con_rows(Xc[i], ss[i * self.nrow:(i + 1) * self.nrow])
        if d:
            pts = [xk] + Xc
            for i in range(1, d + 1):
                sub = at(Xc[i - 1])
                for a in range(nx):
                    R.append(self.dt * rhs[a].subs(sub, simultaneous=True) / self.sx[a] - sum(self.C[j, i] * pts[j][a] for j in range(d + 1)))
            xend = [sum(self.D[j] * pts[j][a] for j in range(d + 1)) for a in range(nx)]
        else:
            sub = at(xk)
            xend = [rhs[a].subs(sub, simultaneous=True) / self.sx[a] for a in range(nx)]
        for a in range(nx):
            R.append(xn[a] - (xend[a] + (wk[a] if self.noise else 0)))
        R += con_rows(xk, ss[d * self.nrow:])
        self.mk = len(R)
        # cost of the interval: arrival at k = 0, stage term at k >= 1 (both on x_k)
        dx = sp.Matrix([self.sx[a] * xk[a] - xa[a] for a in range(nx)])
        cost = first * (dx.T * sp.Matrix(self.Wx) * dx)[0, 0]
        if ne:
            dp = sp.Matrix([self.sp[j] * pe[j] - pa[j] for j in range(ne)])
            cost += first * (dp.T * sp.Matrix(self.Wp) * dp)[0, 0]
        h = [e.subs(at(xk), simultaneous=True) for e in m.meas]
        r = sp.Matrix([h[a] - ym[a] for a in range(ny)])
        cost += inner * (r.T * sp.Matrix(self.Wy) * r)[0, 0]
        if self.noise:
            ws = sp.Matrix([self.sw[a] * wk[a] for a in range(nx)])
            cost += inner * (ws.T * sp.Matrix(self.Ww) * ws)[0, 0]
        q = pe + xk + wk + [s for row in Xc for s in row] + ss + xn
        self.nq = len(q)
        lam = [sp.Symbol(f'l{i}') for i in range(self.mk)]
        args = [q, um, ym, pf, xa, pa, [first, inner], lam]
        self._R = _lam(R, args)
        self._JR = _lam(sp.Matrix(R).jacobian(q).tolist(), args)
        L = cost + sum(l * rr for l, rr in zip(lam, R))
        gL = [sp.diff(L, a) for a in q]
        self._HL = _lam([[sp.diff(gL[i], q[j]) if j >= i else 0 for j in range(len(q))] for i in range(len(q))], args)
        self._cost = _lam([cost], args)
        self._gcost = _lam([sp.diff(cost, a) for a in q], args)
        # ---- reference layout (mhe.py:614-671) ----
        N = self.N
        off = npar
        self.p_ind = [list(range(npar))] if npar else []
        self.x_ind = [list(range(off + k * nx, off + (k + 1) * nx)) for k in range(N + 1)]
        off += (N + 1) * nx
        self.w_ind = [list(range(off + k * nx, off + (k + 1) * nx)) for k in range(N)] if self.noise else []
        off += N * nx if self.noise else 0
        self.ip_ind = [list(range(off + k * d * nx, off + (k + 1) * d * nx)) for k in range(N)] if d else []
        self.n_v = off + N * d * nx
        self.n_g = N * (d * nx + nx + (d + 1) * self.ncon)


class MheGenIpm(DenseIpm):
    """Free variables w = [p_est | x_0..x_N | w_0..w_{N-1} | Xc_0..Xc_{N-1}]."""

    def __init__(self, prob: MheGenProblem, options: IpmOptions | None = None):
        self.pb = pb = prob
        self.o = o = options or IpmOptions()
        N, nx, ne, d = pb.N, pb.nx, pb.ne, pb.d
        self.o_x = ne
        self.o_w = self.o_x + (N + 1) * nx
        self.o_c = self.o_w + (N * nx if pb.noise else 0)
        self.o_s = self.o_c + N * d * nx
        self.ns = (d + 1) * pb.nrow
        self.nw = self.o_s + N * self.ns
        self.m = N * pb.mk
        slb, sub = np.array([r[1] for r in pb.rows] * (d + 1)), np.array([r[2] for r in pb.rows] * (d + 1))
        lb = np.concatenate([pb.p_lb, np.tile(pb.x_lb, N + 1), np.tile(pb.w_lb, N if pb.noise else 0), np.tile(pb.x_lb, N * d), np.tile(slb, N)])
        ub = np.concatenate([pb.p_ub, np.tile(pb.x_ub, N + 1), np.tile(pb.w_ub, N if pb.noise else 0), np.tile(pb.x_ub, N * d), np.tile(sub, N)])
        r = o.bound_relax_factor
        self.lb = np.where(np.isfinite(lb), lb - r * np.maximum(1, np.abs(lb)), lb)
        self.ub = np.where(np.isfinite(ub), ub + r * np.maximum(1, np.abs(ub)), ub)
        self.has_l, self.has_u = np.isfinite(self.lb), np.isfinite(self.ub)

    def qcols(self, k):
        pb = self.pb
        nx, d = pb.nx, pb.d
        c = list(range(pb.ne)) + [self.o_x + k * nx + i for i in range(nx)]
        if pb.noise:
            c += [self.o_w + k * nx + i for i in range(nx)]
        c += list(range(self.o_c + k * d * nx, self.o_c + (k + 1) * d * nx))
        c += list(range(self.o_s + k * self.ns, self.o_s + (k + 1) * self.ns))
        return c + [self.o_x + (k + 1) * nx + i for i in range(nx)]

    def _args(self, w, data, k, lam):
        B = w.shape[0]
        fl = np.tile([1.0 if k == 0 else 0.0, 0.0 if k == 0 else 1.0], (B, 1))
        return w[:, self.qcols(k)], data['u_meas'][:, k], data['y_meas'][:, k], data['p_fixed'], data['x_arrival'], data['p_arrival'], fl, lam

    def eval_fc(self, w, data):
        pb = self.pb
        B = w.shape[0]
        f = np.zeros(B)
        c = np.empty((B, pb.N, pb.mk))
        l0 = np.zeros((B, pb.mk))
        for k in range(pb.N):
            a = self._args(w, data, k, l0)
            c[:, k] = pb._R(*a)
            f += pb._cost(*a)[:, 0]
        return f, c.reshape(B, -1)

    def eval_all(self, w, lam, data):
        pb = self.pb
        B = w.shape[0]
        bi = np.arange(B)
        f = np.zeros(B)
        g = np.zeros((B, self.nw))
        c = np.empty((B, pb.N, pb.mk))
        J = np.zeros((B, self.m, self.nw))
        W = np.zeros((B, self.nw, self.nw))
        lam = lam.reshape(B, pb.N, pb.mk)
        for k in range(pb.N):
            a = self._args(w, data, k, lam[:, k])
            cols = self.qcols(k)
            c[:, k] = pb._R(*a)
            f += pb._cost(*a)[:, 0]
            rows = list(range(k * pb.mk, (k + 1) * pb.mk))
            J[np.ix_(bi, rows, cols)] += pb._JR(*a)
            H = pb._HL(*a)
            H = H + np.triu(H, 1).transpose(0, 2, 1)
            W[np.ix_(bi, cols, cols)] += H
            g[:, cols] += pb._gcost(*a)
        return f, g, c.reshape(B, -1), J, W

    def data(self, x_arrival, p_arrival, p_fixed, u_meas, y_meas):
        pb = self.pb
        xa = np.atleast_2d(np.asarray(x_arrival, dtype=float))
        B = xa.shape[0]
        bc = lambda v, n: np.broadcast_to(np.atleast_2d(np.asarray(v, dtype=float)), (B, n)) if n else np.zeros((B, 0))   # noqa: E731
        return {'x_arrival': xa, 'p_arrival': bc(p_arrival, pb.ne), 'p_fixed': bc(p_fixed, len(pb.fixed)),
                'u_meas': np.asarray(u_meas, dtype=float).reshape(B, pb.N, pb.nu), 'y_meas': np.asarray(y_meas, dtype=float).reshape(B, pb.N, pb.ny)}

    def solve(self, x_arrival, p_arrival, p_fixed, u_meas, y_meas, w0=None, verbose=False):
        """x_arrival [B,nx], p_arrival [B,ne], p_fixed [B,np-ne] (original units), u_meas [B,N,nu], y_meas [B,N,ny]."""
        pb = self.pb
        data = self.data(x_arrival, p_arrival, p_fixed, u_meas, y_meas)
        B = data['x_arrival'].shape[0]
        if w0 is None:
            w0 = np.concatenate([pb.p_guess, np.tile(pb.x_guess, pb.N + 1), np.tile(pb.w_guess, pb.N if pb.noise else 0),
                                 np.tile(pb.x_guess, pb.N * pb.d)])
        w0 = np.broadcast_to(np.atleast_2d(w0)[:, :self.o_s], (B, self.o_s))
        if self.ns:
            # IPOPT: the row slacks start at their rows' values (of the start point pushed into the interior), pushed inside their bounds
            from .nmpc import _push_interior
            w0 = _push_interior(w0, self.lb[:self.o_s], self.ub[:self.o_s], self.o)
            _, c0 = self.eval_fc(np.concatenate([w0, np.zeros((B, pb.N * self.ns))], axis=1), data)
            c0 = c0.reshape(B, pb.N, pb.mk)
            dn = pb.d * pb.nrow
            s0 = np.concatenate([c0[:, :, :dn], c0[:, :, pb.mk - pb.nrow:]], axis=2)
            w0 = np.concatenate([w0, s0.reshape(B, -1)], axis=1)
        res = self.solve_data(data, w0, verbose)
        w = res['w']
        N, nx, d = pb.N, pb.nx, pb.d
        P = w[:, :pb.ne]
        X = w[:, self.o_x:self.o_w].reshape(B, N + 1, nx)
        Wn = w[:, self.o_w:self.o_c].reshape(B, N, nx) if pb.noise else np.zeros((B, N, 0))
        pall = np.zeros((B, pb.np_))
        pall[:, pb.est] = P
        pall[:, pb.fixed] = data['p_fixed']
        res.update(P=P, p_opt=P * pb.sp, X=X, Wn=Wn, x_opt=X[:, -1] * pb.sx, Xc=w[:, self.o_c:self.o_s].reshape(B, N, d, nx),
                   v=np.concatenate([pall, w[:, pb.ne:self.o_s]], axis=1))
        return res

    def lam_g(self, res):
        """Multipliers in the reference's row order: per interval [rows at the collocation points (d x n_con) | collocation rows |
        continuity | rows at the node (n_con)]; rows without a finite bound: 0."""
        pb = self.pb
        B = res['lam'].shape[0]
        d, nrow, nc, nx = pb.d, pb.nrow, pb.ncon, pb.nx
        lam = res['lam'].reshape(B, pb.N, pb.mk)
        out = np.zeros((B, pb.N, d * nc + d * nx + nx + nc))
        for i in range(d):
            for r, (j, _, _) in enumerate(pb.rows):
                out[:, :, i * nc + j] = lam[:, :, i * nrow + r]
        out[:, :, d * nc:d * nc + d * nx + nx] = lam[:, :, d * nrow:d * nrow + d * nx + nx]
        for r, (j, _, _) in enumerate(pb.rows):
            out[:, :, d * nc + d * nx + nx + j] = lam[:, :, d * nrow + d * nx + nx + r]
        return out.reshape(B, -1)

    def w_from_v(self, v):
        pb = self.pb
        v = np.atleast_2d(v)
        return np.concatenate([v[:, pb.est], v[:, pb.np_:]], axis=1)         # (the row slacks always restart at their rows' values)
