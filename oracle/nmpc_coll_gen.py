"""Oracle: the general NMPC transcription under the reference's DEFAULT integration method - direct collocation in its
SIMULTANEOUS form - for an ODE or a semi-explicit DAE, with a path variable, nonlinear stage constraints (hard or soft, one
slack vector shared by all stages) and the continuous or the discrete objective.  This is BASELINE configuration 5 as it is
written (path following on a DAE with soft constraints) on the dense interior-point solver of oracle/nmpc.py.

TEST INFRASTRUCTURE ONLY - never imported by the product package.   PARITY UNPINNED: the reference's DAE test
(tests/test_NMPC.py:1866-1987), its path-following runs (:742-775, path_following_mpc.ipynb) and its soft-constraint runs assert
no number; the interior-point method and the plain collocation transcription this module extends are pinned by the CSTR
notebook (oracle/nmpc_coll.py).  tests/test_oracle_nmpc_coll_gen.py checks this module against oracle/nmpc_coll.py,
oracle/nmpc_dae.py (special cases) and scipy SLSQP on the same NLP.

Restated from hilo_mpc/modules/controller/mpc.py `_setup` with `integration_method='collocation'`:
  * path variable (:1173-1204): theta is appended to the model as a state with the virtual input u_theta, theta' = u_theta
    for a continuous model (:1192) - so it has collocation states like every other state; bounds / guesses / unit scaling
    (:1194-1201); optional (u_theta - u_pf_ref)^2 u_pf_weight in the Lagrange term (:1202-1204); x_0 is pinned for the
    ORIGINAL states only (:785-789).
  * decision vector (:1462-1548):  v = [x_0..x_N | u_0..u_{N-1} | z_0..z_N | (ip_k, zp_k) per interval | e_soft_stage]
    - the slack of a soft stage constraint sits BEHIND the collocation blocks (:1529-1537 follows :1497-1527);
    the node blocks z_0..z_N enter no row and no cost (they stay at the guess).
  * rows per interval (:1338-1372 `gk_col`, then :1657-1669, :1700-1725):
      [ for every collocation point i = 1..d: the stage-constraint residual at (x_{k,i}, u_k, z_{k,i})      (:1339-1356)
      | collocation equations: per point [dt f(x_{k,i}, z_{k,i}, u_k) - sum_j C[j,i] x_{k,j} | g(x_{k,i}, z_{k,i}, u_k)]
      | continuity x_{k+1} - sum_j D_j x_{k,j}
      | the stage-constraint residual at the node (x_k, u_k, zp_k)                                          (:1700-1725) ]
    soft residual: [c - e ; -c - e] <= [ub ; -lb] (:1271-1278); hard: lb <= c <= ub.  The penalty e^T W e is added once per
    interval by the node rows' branch only (:1708), W = 1e4 I by default (modeling.py:875).
  * the node residual receives `zp[ii, 0]` for the algebraic state (:1707): the WHOLE block of degree * n_z collocation values
    where the function expects n_z.  For degree = 1 that is z at the single collocation point (Radau: the END of the interval)
    and this module restates it as it is; for degree > 1 and n_z > 0 CasADi rejects the call (shape mismatch), i.e. the
    reference cannot build that NLP at all.  There the node residual is evaluated with the algebraic state CONSISTENT with the
    node, g(x_k, z, u_k) = 0 - an extension where the reference has no behaviour (stated in DESIGN.md 7); it needs an
    algebraic equation sympy can solve for z.
  * objective (:1676-1682, optimizer.py:1423-1426): 'continuous' (default for a continuous model) integrates the Lagrange term
    with the collocation quadrature dt sum_i B_i l(x_{k,i}, u_k) (modeling.py:1195); 'discrete' sums l(x_k, u_k).  The Mayer
    term acts on the end state; here it is written on the variable x_N (equal on the feasible set), `lam_g` converts the
    multiplier of the last continuity row to the reference's convention.
Constraint expressions act on UN-scaled variables (modeling.py:843-849), costs on the scaled ones.
Inequality rows the IPOPT way: a slack s per row, d(w) - s = 0, bounds on s relaxed by bound_relax_factor.  Rows without a
finite bound are dropped; `lam_g` returns zeros for them.
"""
from __future__ import annotations

import numpy as np
import sympy as sp

from .models import _lam
from .nmpc import DenseIpm, IpmOptions, NmpcProblem, _push_interior, _wmat
from .nmpc_coll import polynomial_basis

INF = np.inf


def _parse(expr, names):
    return sp.sympify(expr, locals=names) if isinstance(expr, str) else sp.sympify(expr)


class GenCollProblem(NmpcProblem):
    """path / constraint: the dictionaries of oracle/nmpc_gen.py::GenNmpcProblem (constraint expressions may also name the
    algebraic states and the path variable); generic_stage: sympy expression of the model's symbols (scaled variables, the
    quirk restated in oracle/nmpc_coll.py)."""

    def __init__(self, model, dt, N, degree=3, points='radau', objective='continuous', path=None, constraint=None,
                 generic_stage=None, z_guess=None, z_lb=None, z_ub=None, terminal=None, min_time=None, **kw):
        kw.pop('order', None)
        super().__init__(model, dt, N, **kw)
        assert not model.discrete and objective in ('continuous', 'discrete')
        m = model
        self.objective, self.d, self.path = objective, degree, path
        d = degree
        self.B, self.C, self.D, self.tau = polynomial_basis(degree, points)
        nx, nu = self.nx, self.nu
        self.nth = nth = 1 if path else 0
        self.nxa, self.nua, self.nzalg = nx + nth, nu + nth, len(m.z)
        nxa, nua, nzg = self.nxa, self.nua, self.nzalg
        self.z_guess = np.zeros(nzg) if z_guess is None else np.asarray(z_guess, dtype=float).reshape(nzg)
        # box of the algebraic states: the bounds of the zp blocks of v (mpc.py:645-701, :1512-1518; the node blocks enter nothing)
        self.z_lb = np.full(nzg, -INF) if z_lb is None else np.broadcast_to(np.asarray(z_lb, dtype=float), (nzg,)).copy()
        self.z_ub = np.full(nzg, INF) if z_ub is None else np.broadcast_to(np.asarray(z_ub, dtype=float), (nzg,)).copy()
        self.sxa = np.concatenate([self.sx, np.ones(nth)])
        self.sua = np.concatenate([self.su, np.ones(nth)])
        if path:
            self.x_lb = np.concatenate([self.x_lb, [path.get('theta_lb', 0.)]])
            self.x_ub = np.concatenate([self.x_ub, [path.get('theta_ub', INF)]])
            self.u_lb = np.concatenate([self.u_lb, [path.get('u_pf_lb', 1e-4)]])
            self.u_ub = np.concatenate([self.u_ub, [path.get('u_pf_ub', 1.)]])
            self.x_guess = np.concatenate([self.x_guess, [path.get('theta_guess', 0.)]])
            self.u_guess = np.concatenate([self.u_guess, [path.get('u_pf_lb', 1e-4) + 1e-4]])
        # ---- symbols of one interval (scaled variables) ----
        xk = [sp.Symbol(f'xk{i}') for i in range(nxa)]
        uk = [sp.Symbol(f'uk{i}') for i in range(nua)]
        Xc = [[sp.Symbol(f'xc{i}_{a}') for a in range(nxa)] for i in range(d)]
        Zc = [[sp.Symbol(f'zc{i}_{a}') for a in range(nzg)] for i in range(d)]
        uo = [sp.Symbol(f'uold{i}') for i in range(nu)]
        f0 = sp.Symbol('first_interval')          # 1 in interval 0 (the input-change term, mpc.py:1631-1635), else 0
        th = sp.Symbol(path.get('name', 'theta')) if path else None
        names = {str(s): s for s in m.x + m.u + m.z}
        if th is not None:
            names[str(th)] = th

        def at(xs, us, zs):
            """substitution of the model's symbols by the interval's (un-scaled values of scaled variables)"""
            sub = {m.x[a]: self.sx[a] * xs[a] for a in range(nx)}
            sub.update({m.u[a]: self.su[a] * us[a] for a in range(nu)})
            if zs is not None:
                sub.update({m.z[a]: zs[a] for a in range(nzg)})
            if th is not None:
                sub[th] = xs[nx]
            return sub

        def lagrange(xs, us):
            """Lagrange term on the scaled variables of a point"""
            zz = list(xs[:nx]) + list(us[:nu])
            dz = sp.Matrix([zz[i] - self.zref[i] for i in range(self.nz)])
            l = (dz.T * sp.Matrix(self.Wz) * dz)[0, 0]
            du = sp.Matrix([us[i] - uo[i] for i in range(nu)])
            l += f0 * (du.T * sp.Matrix(self.Wdu) * du)[0, 0]
            if path:
                for ind, W, refs in path.get('stage', []):
                    W = _wmat(W, len(ind))
                    dd = sp.Matrix([xs[i] - _parse(r, {str(th): th}).subs(th, xs[nx]) for i, r in zip(ind, refs)])
                    l += (dd.T * sp.Matrix(W) * dd)[0, 0]
                if path.get('u_pf_ref') is not None:
                    l += (us[nu] - path['u_pf_ref']) ** 2 * path.get('u_pf_weight', 10.)
            if generic_stage is not None:
                sub = {s: xs[i] for i, s in enumerate(m.x)}
                sub.update({s: us[i] for i, s in enumerate(m.u)})
                l += sp.sympify(generic_stage).subs(sub, simultaneous=True)
            return l

        # ---- constraint rows ----
        self.ne = 0
        self.rows = []          # (expression index, sign, slack index or -1, lb, ub, position in the reference's residual)
        cexpr = []
        if constraint:
            cexpr = [_parse(e, names) for e in constraint['expr']]
            nc = len(cexpr)
            lb = np.broadcast_to(np.asarray(constraint.get('lb', -INF), dtype=float), (nc,))
            ub = np.broadcast_to(np.asarray(constraint.get('ub', INF), dtype=float), (nc,))
            if constraint.get('soft'):
                self.ne = nc
                W = constraint.get('weight')
                self.We = np.diag(np.ones(nc) * 1e4) if W is None else _wmat(W, nc)             # modeling.py:875
                self.e_ub = np.broadcast_to(np.asarray(constraint.get('max_violation', INF), dtype=float), (nc,))
                for j in range(nc):
                    if np.isfinite(ub[j]):
                        self.rows.append((j, 1., j, -INF, ub[j], j))
                    if np.isfinite(lb[j]):
                        self.rows.append((j, -1., j, -INF, -lb[j], nc + j))
            else:
                for j in range(nc):
                    if np.isfinite(lb[j]) or np.isfinite(ub[j]):
                        self.rows.append((j, 1., -1, lb[j], ub[j], j))
            self.n_con_ref = 2 * nc if constraint.get('soft') else nc
        else:
            self.n_con_ref = 0
        self.nrow = nrow = len(self.rows)
        ne = self.ne
        es = [sp.Symbol(f'e{j}') for j in range(ne)]
        ss = [sp.Symbol(f's{r}') for r in range((d + 1) * nrow)]
        xk1 = [sp.Symbol(f'xn{i}') for i in range(nxa)]
        uses_z = any(e.has(*m.z) for e in cexpr) if nzg else False
        # algebraic state at the node (see the module docstring)
        znode = None
        if nzg and uses_z:
            if d == 1:
                znode = Zc[0]
            else:
                sol = sp.solve(m.alg, m.z, dict=True)
                if len(sol) != 1:
                    raise NotImplementedError("the node rows need an algebraic equation with one explicit solution for z")
                znode = [sol[0][s].subs(at(xk, uk, None), simultaneous=True) for s in m.z]

        def con_rows(xs, us, zs, sl):
            sub = at(xs, us, zs)
            cv = [e.subs(sub, simultaneous=True) for e in cexpr]
            out = []
            for r, (j, sg, ei, _, _, _) in enumerate(self.rows):
                v = sg * cv[j] - sl[r]
                if ei >= 0:
                    v = v - es[ei]
                out.append(v)
            return out

        # minimum-time problems (mpc.py:859-866, :1452-1453, :1606-1617, :1746-1754): the N sampling intervals are variables
        # (the last block of v, bounds [0, inf), guess dt), forced equal by N - 1 rows dt_k - dt_{k+1} = 0 at the end of g;
        # J += weight * sum(dt).  min_time = weight.
        self.min_time = None if min_time is None else float(min_time)
        dts = sp.Symbol('dtk') if self.min_time is not None else self.dt
        self._dts = dts
        pts = [xk] + Xc
        R = []
        for i in range(1, d + 1):                                    # A: residuals at the collocation points
            R += con_rows(Xc[i - 1], uk, Zc[i - 1] if nzg else None, ss[(i - 1) * nrow:i * nrow])
        for i in range(1, d + 1):                                    # B: collocation equations
            sub = at(Xc[i - 1], uk, Zc[i - 1] if nzg else None)
            f = [e.subs(sub, simultaneous=True) / self.sx[a] for a, e in enumerate(m.ode)]       # base.py:1562-1591
            if nth:
                f.append(uk[nu])                                                                 # theta' = u_theta (mpc.py:1192)
            for a in range(nxa):
                R.append(dts * f[a] - sum(self.C[j, i] * pts[j][a] for j in range(d + 1)))
            R += [e.subs(sub, simultaneous=True) for e in m.alg]
        for a in range(nxa):                                         # C: continuity
            R.append(xk1[a] - sum(self.D[j] * pts[j][a] for j in range(d + 1)))
        R += con_rows(xk, uk, znode, ss[d * nrow:])                  # D: residuals at the node
        self.mk = len(R)
        lam = [sp.Symbol(f'l{r}') for r in range(self.mk)]
        if objective == 'continuous':
            cost = sum(dts * self.B[i] * lagrange(Xc[i - 1], uk) for i in range(1, d + 1))
        else:
            cost = lagrange(xk, uk)
        if self.min_time is not None:
            cost = cost + self.min_time * dts
        qn = xk + uk + [s for row in Xc for s in row] + [s for row in Zc for s in row]          # enter non-linearly
        if self.min_time is not None:
            qn = qn + [dts]
        ql = es + ss + xk1                                                                      # enter linearly (rows); e^T W e is added apart
        self.nqn, self.nql = len(qn), len(ql)
        q = qn + ql
        args = [q, m.p, uo, [f0], lam]
        self._R = _lam(R, args)
        self._JR = _lam(sp.Matrix(R).jacobian(q).tolist(), args)
        L = cost + sum(l * r for l, r in zip(lam, R))
        gL = [sp.diff(L, a) for a in qn]
        self._HL = _lam([[sp.diff(gL[i], qn[j]) if j >= i else 0 for j in range(len(qn))] for i in range(len(qn))], args)
        self._cost = _lam([cost], args)
        self._gcost = _lam([sp.diff(cost, a) for a in qn], args)
        # ---- terminal cost on x_N ----
        xN = xk
        dN = sp.Matrix([xN[i] - self.xrefN[i] for i in range(nx)])
        V = (dN.T * sp.Matrix(self.WN) * dN)[0, 0]
        if path:
            for ind, W, refs in path.get('terminal', []):
                W = _wmat(W, len(ind))
                dd = sp.Matrix([xN[i] - _parse(r, {str(th): th}).subs(th, xN[nx]) for i, r in zip(ind, refs)])
                V += (dd.T * sp.Matrix(W) * dd)[0, 0]
        self._V = _lam([V], [xN])
        self._gV = _lam([sp.diff(V, a) for a in xN], [xN])
        self._HV = _lam([[sp.diff(V, a, b) for b in xN] for a in xN], [xN])
        # ---- hard terminal constraint (mpc.py:1693-1700): rows on the integrated end state x_end of the last interval, between its
        # continuity rows and its node rows in g.  Restated on the variable x_N (x_N = x_end at every feasible point) like the Mayer
        # term above: the multiplier of the last continuity row is handed out in the reference's convention, lambda + grad V +
        # (grad c_t)' nu_t (GenCollIpm.lam_g).  terminal = dict(expr=[...], lb, ub) in the model's (un-scaled) states.
        self.trows = []                                              # (expression index, lb, ub)
        self.n_tcon_ref = 0
        if terminal:
            tex = [_parse(e, names) for e in terminal['expr']]
            self.n_tcon_ref = len(tex)
            tlb = np.broadcast_to(np.asarray(terminal.get('lb', -INF), dtype=float), (len(tex),))
            tub = np.broadcast_to(np.asarray(terminal.get('ub', INF), dtype=float), (len(tex),))
            self.trows = [(j, tlb[j], tub[j]) for j in range(len(tex)) if np.isfinite(tlb[j]) or np.isfinite(tub[j])]
            subN = {m.x[a]: self.sx[a] * xN[a] for a in range(nx)}
            T = [tex[j].subs(subN, simultaneous=True) for j, _, _ in self.trows]
            lt = [sp.Symbol(f'lt{r}') for r in range(len(T))]
            self._T = _lam(T, [xN, m.p])
            self._JT = _lam(sp.Matrix(T).jacobian(xN).tolist(), [xN, m.p])
            LT = sum(a * b for a, b in zip(lt, T))
            self._HT = _lam([[sp.diff(LT, a, b) for b in xN] for a in xN], [xN, m.p, lt])
        self.ntrow = len(self.trows)
        # ---- reference layout (mpc.py:1462-1548) ----
        off = (N + 1) * nxa
        self.x_ind = [list(range(k * nxa, (k + 1) * nxa)) for k in range(N + 1)]
        self.u_ind = [list(range(off + k * nua, off + (k + 1) * nua)) for k in range(N)]
        off += N * nua
        self.z_ind = [list(range(off + k * nzg, off + (k + 1) * nzg)) for k in range(N + 1)] if nzg else []
        off += (N + 1) * nzg
        self.ip_ind, self.zp_ind = [], []
        for k in range(N):
            self.ip_ind.append(list(range(off, off + d * nxa)))
            off += d * nxa
            if nzg:
                self.zp_ind.append(list(range(off, off + d * nzg)))
                off += d * nzg
        self.e_ind = list(range(off, off + ne))
        self.n_v = off + ne
        self.dt_ind = list(range(self.n_v, self.n_v + N)) if self.min_time is not None else []
        self.n_v += len(self.dt_ind)
        self.n_g = N * (d * self.n_con_ref + d * (nxa + nzg) + nxa + self.n_con_ref) + self.n_tcon_ref + max(0, len(self.dt_ind) - 1)


class GenCollIpm(DenseIpm):
    """Free variables w = [theta_0 | xa_1..xa_N | ua_0..ua_{N-1} | (Xc_k, Zc_k) per interval | e | s_k per interval]."""

    def __init__(self, prob: GenCollProblem, options: IpmOptions | None = None):
        self.pb = pb = prob
        self.o = o = options or IpmOptions()
        N, nxa, nua, d, nzg, nth, ne, nrow = pb.N, pb.nxa, pb.nua, pb.d, pb.nzalg, pb.nth, pb.ne, pb.nrow
        self.o_x = nth
        self.o_u = self.o_x + N * nxa
        self.o_c = self.o_u + N * nua
        self.blk = d * (nxa + nzg)
        self.o_e = self.o_c + N * self.blk
        self.o_s = self.o_e + ne
        self.ns = (d + 1) * nrow
        self.o_t = self.o_s + N * self.ns                            # slacks of the terminal rows
        self.o_dt = self.o_t + pb.ntrow                              # sampling intervals (minimum-time problems)
        self.ndt = N if pb.min_time is not None else 0
        self.nw = self.o_dt + self.ndt
        self.mk = pb.mk
        self.m = N * self.mk + pb.ntrow + max(0, self.ndt - 1)
        zl, zu = pb.z_lb, pb.z_ub
        blk_lb = np.concatenate([np.tile(pb.x_lb, d), np.tile(zl, d)])
        blk_ub = np.concatenate([np.tile(pb.x_ub, d), np.tile(zu, d)])
        slb = np.array([r[3] for r in pb.rows] * (d + 1))
        sub = np.array([r[4] for r in pb.rows] * (d + 1))
        lb = np.concatenate([pb.x_lb[pb.nx:], np.tile(pb.x_lb, N), np.tile(pb.u_lb, N), np.tile(blk_lb, N), np.zeros(ne), np.tile(slb, N),
                             np.array([r[1] for r in pb.trows]), np.zeros(self.ndt)])
        ub = np.concatenate([pb.x_ub[pb.nx:], np.tile(pb.x_ub, N), np.tile(pb.u_ub, N), np.tile(blk_ub, N),
                             pb.e_ub if ne else np.zeros(0), np.tile(sub, N), np.array([r[2] for r in pb.trows]), np.full(self.ndt, np.inf)])
        r = o.bound_relax_factor
        self.lb = np.where(np.isfinite(lb), lb - r * np.maximum(1, np.abs(lb)), lb)
        self.ub = np.where(np.isfinite(ub), ub + r * np.maximum(1, np.abs(ub)), ub)
        self.has_l, self.has_u = np.isfinite(self.lb), np.isfinite(self.ub)

    # columns of the interval's q = [xk | uk | Xc | Zc | e | s_k | x_{k+1}] in w (-1: the pinned part of x_0)
    def qcols(self, k):
        pb = self.pb
        if k == 0:
            cx = [-1] * pb.nx + list(range(pb.nth))
        else:
            cx = [self.o_x + (k - 1) * pb.nxa + i for i in range(pb.nxa)]
        cu = [self.o_u + k * pb.nua + i for i in range(pb.nua)]
        cb = list(range(self.o_c + k * self.blk, self.o_c + (k + 1) * self.blk))
        if self.ndt:
            cb = cb + [self.o_dt + k]
        ce = list(range(self.o_e, self.o_s))
        cs = list(range(self.o_s + k * self.ns, self.o_s + (k + 1) * self.ns))
        cn = [self.o_x + k * pb.nxa + i for i in range(pb.nxa)]
        return cx + cu + cb + ce + cs + cn

    def _unpack(self, w, x0):
        pb = self.pb
        B, N, nxa, nua, d, nzg = w.shape[0], pb.N, pb.nxa, pb.nua, pb.d, pb.nzalg
        X = np.empty((B, N + 1, nxa))
        X[:, 0, :pb.nx] = x0
        X[:, 0, pb.nx:] = w[:, :pb.nth]
        X[:, 1:] = w[:, self.o_x:self.o_u].reshape(B, N, nxa)
        U = w[:, self.o_u:self.o_c].reshape(B, N, nua)
        blk = w[:, self.o_c:self.o_e].reshape(B, N, self.blk)
        E = w[:, self.o_e:self.o_s]
        S = w[:, self.o_s:self.o_t].reshape(B, N, self.ns)
        self._dtv = w[:, self.o_dt:self.o_dt + self.ndt]
        return X, U, blk, E, S

    def _q(self, X, U, blk, E, S, k):
        if self.ndt:
            return np.concatenate([X[:, k], U[:, k], blk[:, k], self._dtv[:, k:k + 1], E, S[:, k], X[:, k + 1]], axis=1)
        return np.concatenate([X[:, k], U[:, k], blk[:, k], E, S[:, k], X[:, k + 1]], axis=1)

    def _args(self, q, data, k, lam):
        B = q.shape[0]
        uo = data.get('u_old')
        uo = np.zeros((B, self.pb.nu)) if uo is None else uo
        f0 = np.full((B, 1), 1.0 if (k == 0 and data.get('u_old') is not None) else 0.0)
        return q, data['p'], uo, f0, lam

    def eval_fc(self, w, data):
        pb = self.pb
        X, U, blk, E, S = self._unpack(w, data['x0'])
        B, N = w.shape[0], pb.N
        f = np.zeros(B)
        c = np.empty((B, N, self.mk))
        l0 = np.zeros((B, self.mk))
        for k in range(N):
            a = self._args(self._q(X, U, blk, E, S, k), data, k, l0)
            c[:, k] = pb._R(*a)
            f += pb._cost(*a)[:, 0]
            if pb.ne:
                f += np.einsum('bi,ij,bj->b', E, pb.We, E)                              # mpc.py:1708: once per interval
        c = c.reshape(B, -1)
        if pb.ntrow:
            c = np.concatenate([c, pb._T(X[:, N], data['p']) - w[:, self.o_t:self.o_dt]], axis=1)
        if self.ndt > 1:
            c = np.concatenate([c, self._dtv[:, :-1] - self._dtv[:, 1:]], axis=1)
        return f + pb._V(X[:, N])[:, 0], c

    def eval_all(self, w, lam, data):
        pb = self.pb
        X, U, blk, E, S = self._unpack(w, data['x0'])
        B, N, nqn = w.shape[0], pb.N, pb.nqn
        bi = np.arange(B)
        f = np.zeros(B)
        g = np.zeros((B, self.nw))
        c = np.empty((B, N, self.mk))
        J = np.zeros((B, self.m, self.nw))
        W = np.zeros((B, self.nw, self.nw))
        lamT = lam[:, N * self.mk:N * self.mk + pb.ntrow]
        lam = lam[:, :N * self.mk].reshape(B, N, self.mk)
        ecols = list(range(self.o_e, self.o_s))
        for k in range(N):
            a = self._args(self._q(X, U, blk, E, S, k), data, k, lam[:, k])
            cols = self.qcols(k)
            keep = [j for j, cix in enumerate(cols) if cix >= 0]
            kc = [cols[j] for j in keep]
            c[:, k] = pb._R(*a)
            f += pb._cost(*a)[:, 0]
            rows = list(range(k * self.mk, (k + 1) * self.mk))
            J[np.ix_(bi, rows, kc)] += pb._JR(*a)[:, :, keep]
            H = pb._HL(*a)
            H = H + np.triu(H, 1).transpose(0, 2, 1)
            kn = [j for j in keep if j < nqn]
            kcn = [cols[j] for j in kn]
            W[np.ix_(bi, kcn, kcn)] += H[np.ix_(bi, kn, kn)]
            g[:, kcn] += pb._gcost(*a)[:, kn]
            if pb.ne:
                f += np.einsum('bi,ij,bj->b', E, pb.We, E)
                g[:, ecols] += E @ (pb.We + pb.We.T)
                W[np.ix_(bi, ecols, ecols)] += pb.We + pb.We.T
        xi = [self.o_x + (N - 1) * pb.nxa + i for i in range(pb.nxa)]
        f += pb._V(X[:, N])[:, 0]
        g[:, xi] += pb._gV(X[:, N])
        W[np.ix_(bi, xi, xi)] += pb._HV(X[:, N])
        c = c.reshape(B, -1)
        if pb.ntrow:
            trows = list(range(N * self.mk, N * self.mk + pb.ntrow))
            c = np.concatenate([c, pb._T(X[:, N], data['p']) - w[:, self.o_t:self.o_dt]], axis=1)
            J[np.ix_(bi, trows, xi)] += pb._JT(X[:, N], data['p'])
            for r in range(pb.ntrow):
                J[:, N * self.mk + r, self.o_t + r] = -1.0
            W[np.ix_(bi, xi, xi)] += pb._HT(X[:, N], data['p'], lamT)
        if self.ndt > 1:
            c = np.concatenate([c, self._dtv[:, :-1] - self._dtv[:, 1:]], axis=1)
            r0 = N * self.mk + pb.ntrow
            for k in range(N - 1):
                J[:, r0 + k, self.o_dt + k] = 1.0
                J[:, r0 + k, self.o_dt + k + 1] = -1.0
        return f, g, c, J, W

    def start(self, x0, data):
        """w_0 of the reference's guess (mpc.py:1468-1537): states / inputs / collocation blocks tiled, slacks of the soft
        constraint 0; the row slacks start at their rows' values (IPOPT's slack initialisation), pushed into the interior."""
        pb, o = self.pb, self.o
        B = x0.shape[0]
        blk0 = np.concatenate([np.tile(pb.x_guess, pb.d), np.tile(pb.z_guess, pb.d)])
        w0 = np.concatenate([pb.x_guess[pb.nx:], np.tile(pb.x_guess, pb.N), np.tile(pb.u_guess, pb.N), np.tile(blk0, pb.N), np.zeros(pb.ne)])
        w0 = np.broadcast_to(w0, (B, self.o_s))
        return w0

    def _tail(self, B):
        """the variables behind the row slacks that have a guess of their own: the sampling intervals (v_guess = dt, mpc.py:1614)"""
        return np.full((B, self.ndt), self.pb.dt)

    def _with_slacks(self, w0, data):
        pb, o = self.pb, self.o
        B = w0.shape[0]
        w0 = _push_interior(w0, self.lb[:self.o_s], self.ub[:self.o_s], o)
        tail = _push_interior(self._tail(B), self.lb[self.o_dt:], self.ub[self.o_dt:], o) if self.ndt else np.zeros((B, 0))
        if not pb.nrow and not pb.ntrow:
            return np.concatenate([w0, tail], axis=1)
        wz = np.concatenate([w0, np.zeros((B, pb.N * self.ns + pb.ntrow)), tail], axis=1)
        _, c = self.eval_fc(wz, data)
        cT = c[:, pb.N * self.mk:pb.N * self.mk + pb.ntrow]
        c = c[:, :pb.N * self.mk].reshape(B, pb.N, self.mk)
        d, nrow = pb.d, pb.nrow
        s0 = np.concatenate([c[:, :, :d * nrow], c[:, :, self.mk - nrow:]], axis=2)         # rows = d(w) - s with s = 0
        return np.concatenate([w0, s0.reshape(B, -1), cT, tail], axis=1)

    def solve(self, x0, p, w0=None, u_old=None, verbose=False):
        pb = self.pb
        x0 = np.atleast_2d(np.asarray(x0, dtype=float)) / pb.sx
        B = x0.shape[0]
        p = np.broadcast_to(np.atleast_2d(np.asarray(p, dtype=float)), (B, pb.np_)) if pb.np_ else np.zeros((B, 0))
        data = {'x0': x0, 'p': p}
        if u_old is not None:
            data['u_old'] = np.broadcast_to(np.atleast_2d(np.asarray(u_old, dtype=float)), (B, pb.nu))
        w0 = self.start(x0, data) if w0 is None else np.broadcast_to(np.atleast_2d(w0)[:, :self.o_s], (B, self.o_s))
        res = self.solve_data(data, self._with_slacks(w0, data), verbose)
        X, U, blk, E, S = self._unpack(res['w'], x0)
        d, nxa, nzg = pb.d, pb.nxa, pb.nzalg
        res['p_data'] = p
        if self.ndt:
            res['dt'] = res['w'][:, self.o_dt:]
        res.update(X=X, U=U, E=E, S=S, Xc=blk[:, :, :d * nxa].reshape(B, pb.N, d, nxa),
                   Zc=blk[:, :, d * nxa:].reshape(B, pb.N, d, nzg), u0=U[:, 0, :pb.nu] * pb.su, x0=x0)
        return res

    # ---- reference layouts ------------------------------------------------------------------------------------------
    def to_v(self, res):
        pb = self.pb
        B = res['X'].shape[0]
        parts = [res['X'].reshape(B, -1), res['U'].reshape(B, -1)]
        if pb.nzalg:
            parts.append(np.tile(pb.z_guess, (B, pb.N + 1)))
        for k in range(pb.N):
            parts.append(res['Xc'][:, k].reshape(B, -1))
            if pb.nzalg:
                parts.append(res['Zc'][:, k].reshape(B, -1))
        parts.append(res['E'])
        if self.ndt:
            parts.append(res['w'][:, self.o_dt:])
        return np.concatenate(parts, axis=1)

    def w_from_v(self, v):
        """[theta_0 | xa_1.. | ua | blocks | e] from the reference's decision vector"""
        pb = self.pb
        v = np.atleast_2d(v)
        nX = (pb.N + 1) * pb.nxa
        head = [v[:, pb.nx:pb.nxa], v[:, pb.nxa:nX], v[:, nX:nX + pb.N * pb.nua]]
        off = nX + pb.N * pb.nua + (pb.N + 1) * pb.nzalg
        return np.concatenate(head + [v[:, off:]], axis=1)

    def lam_g(self, res):
        """Multipliers in the reference's row order (dropped rows: 0); the last continuity row in the reference's convention
        (Mayer term on the integrated end state, mpc.py:1682): lambda + grad V(x_N)."""
        pb = self.pb
        B = res['lam'].shape[0]
        d, nrow, ncr, nxa, nzg = pb.d, pb.nrow, pb.n_con_ref, pb.nxa, pb.nzalg
        lamT = res['lam'][:, pb.N * self.mk:pb.N * self.mk + pb.ntrow]
        lamD = res['lam'][:, pb.N * self.mk + pb.ntrow:]
        lam = res['lam'][:, :pb.N * self.mk].reshape(B, pb.N, self.mk)
        per = d * ncr + d * (nxa + nzg) + nxa + ncr
        out = np.zeros((B, pb.N, per))
        for i in range(d):
            for r, row in enumerate(pb.rows):
                out[:, :, i * ncr + row[5]] = lam[:, :, i * nrow + r]
        nb = d * (nxa + nzg) + nxa
        out[:, :, d * ncr:d * ncr + nb] = lam[:, :, d * nrow:d * nrow + nb]
        for r, row in enumerate(pb.rows):
            out[:, :, d * ncr + nb + row[5]] = lam[:, :, d * nrow + nb + r]
        out[:, -1, d * ncr + d * (nxa + nzg):d * ncr + nb] += pb._gV(res['X'][:, pb.N])
        if not pb.n_tcon_ref:
            return np.concatenate([out.reshape(B, -1), lamD], axis=1)
        # terminal rows: between the continuity rows and the node rows of the last interval; their pull on x_end joins the
        # multiplier of that interval's continuity rows
        p = res.get('p_data')
        JT = pb._JT(res['X'][:, pb.N], p)
        out[:, -1, d * ncr + d * (nxa + nzg):d * ncr + nb] += np.einsum('br,bri->bi', lamT, JT)
        lt = np.zeros((B, pb.n_tcon_ref))
        for r, (j, _, _) in enumerate(pb.trows):
            lt[:, j] = lamT[:, r]
        flat = out.reshape(B, -1)
        cut = (pb.N - 1) * per + d * ncr + nb
        return np.concatenate([flat[:, :cut], lt, flat[:, cut:], lamD], axis=1)
