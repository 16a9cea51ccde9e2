"""Oracle: the general NMPC transcription - path following and nonlinear stage constraints - on the dense
interior-point solver of oracle/nmpc.py.

TEST INFRASTRUCTURE ONLY - never imported by the product package.   PARITY UNPINNED (see oracle/nmpc.py): the
reference's path-following and soft-constraint runs (tests/test_NMPC.py:742-775, path_following_mpc.ipynb) assert no
numbers, and CasADi/IPOPT cannot be installed here.

Restated from hilo_mpc/modules/controller/mpc.py (pre-discretised model + `integration_method='discrete'`, Q18):
  * path following (:1025-1053, :1173-1204): every path variable theta becomes a model state with its own virtual input,
    theta+ = theta + dt u_theta for a discrete model (explicit Euler, :1188-1191); bounds theta in [theta_lb, theta_ub],
    u_theta in [u_pf_lb, u_pf_ub]; guesses theta_guess and u_pf_lb + 1e-4 (:1194-1195); unit scaling (:1200-1201);
    optional (u_theta - u_pf_ref)^2 u_pf_weight stage term (:1202-1204).  `optimize` pins only the ORIGINAL states of
    x_0 (:785-789): theta_0 is a free, bounded variable.  The path cost substitutes the expression of theta for the
    reference, (s - r(theta))^T W (s - r(theta)) (hilo_mpc/util/modeling.py:252-283), stage and terminal.
  * stage constraints (`GenericConstraint`, modeling.py:820-1005; mpc.py:1271-1283, :1700-1725): rows
    lb <= c(x_k, u_k) <= ub for k = 0..N-1 on UN-scaled variables (modeling.py:843-849); soft: ONE slack vector e >= 0
    shared by all stages (mpc.py:1529-1537), rows c - e <= ub and -c - e <= -lb (:1276-1277), penalty e^T W e added
    once per stage (:1708), W = 1e4 I by default (modeling.py:875), e <= max_violation.
    Rows whose bound is infinite on both sides (e.g. -c - e <= +inf when lb = -inf) constrain nothing and are dropped.
  * decision vector v = [x_0..x_N | u_0..u_{N-1} | e] with the path states/inputs inside x and u (mpc.py:1462-1537);
    g interleaves per stage the shooting defect and the constraint rows (:1667, :1707-1725).
Inequality rows are handled the way IPOPT does (Waechter & Biegler 2006, sec. 3.4 of the implementation paper): a slack
s with d(w) - s = 0 and bounds d_L <= s <= d_U (relaxed by bound_relax_factor), s_0 = d(w_0) pushed into the interior.
"""
from __future__ import annotations

import numpy as np
import sympy as sp

from .models import _lam
from .nmpc import DenseIpm, IpmOptions, NmpcProblem, _push_interior, _wmat

INF = np.inf


def _parse(expr, names):
    return sp.sympify(expr, locals=names) if isinstance(expr, str) else sp.sympify(expr)


class GenNmpcProblem(NmpcProblem):
    """path = dict(name='theta', theta_guess=0., theta_lb=0., theta_ub=inf, u_pf_lb=1e-4, u_pf_ub=1., u_pf_ref=None,
                   u_pf_weight=10., stage=[(state indices, weights, [expr in theta, ...])], terminal=[...])
       constraint = dict(expr=[expr in state/input names, ...], lb=[...], ub=[...], soft=False, weight=None,
                         max_violation=inf)
       generic_stage = sympy expression of the model's state / input symbols added to every stage's cost"""

    def __init__(self, model, dt, N, path=None, constraint=None, terminal_constraint=None, generic_stage=None, custom=None, **kw):
        """custom = dict(fun=f(v, x_ind, u_ind) -> expression(s) of the entries of the (scaled) decision vector v, lb=[...], ub=[...]):
        `set_custom_constraints_function` (optimizer.py:1180-1208) - rows lb <= fun(v, x_ind, u_ind) <= ub appended to g
        (mpc.py:1741-1744), kept here as what they are: DENSE rows over the whole vector (sympy derivatives in v).
        custom['soft'] (mpc.py:1551-1556, :1731-1740): ONE slack e_cus per row behind the other slacks in v, in [0, max_violation],
        start value = the number of rows (:1555), 1e4 e_cus^T e_cus in J, and TWO rows per function: fun - e_cus in (-inf, ub],
        fun + e_cus in [lb, inf)."""
        super().__init__(model, dt, N, **kw)
        nx, nu = self.nx, self.nu
        self.path = path
        self.nth = 1 if path else 0
        self.nxa, self.nua = nx + self.nth, nu + self.nth
        self.nza = self.nxa + self.nua
        names = {str(s): s for s in model.x + model.u}
        zs = [sp.Symbol(f'zs_{i}') for i in range(self.nza)]                 # scaled stage variables
        xs_sym, us_sym = zs[:self.nxa], zs[self.nxa:]
        self.sza = np.concatenate([self.sx, np.ones(self.nth), self.su, np.ones(self.nth)])
        # ---- quadratic part on the augmented z (zero rows for the path variable) ----
        idx = list(range(nx)) + list(range(self.nxa, self.nxa + nu))
        self.Wza = np.zeros((self.nza, self.nza))
        self.Wza[np.ix_(idx, idx)] = self.Wz
        self.zrefa = np.zeros(self.nza)
        self.zrefa[idx] = self.zref
        self.WNa = np.zeros((self.nxa, self.nxa))
        self.WNa[:nx, :nx] = self.WN
        self.xrefNa = np.zeros(self.nxa)
        self.xrefNa[:nx] = self.xrefN
        # ---- path terms (symbolic in the scaled stage variables) ----
        lp, Vp = sp.Integer(0), sp.Integer(0)
        th = None
        if path:
            th = sp.Symbol(path.get('name', 'theta'))
            th_s, uth_s = xs_sym[nx], us_sym[nu]
            for ind, W, refs in path.get('stage', []):
                W = _wmat(W, len(ind))
                d = sp.Matrix([xs_sym[i] - _parse(r, {str(th): th}).subs(th, th_s) for i, r in zip(ind, refs)])
                lp += (d.T * sp.Matrix(W) * d)[0, 0]
            for ind, W, refs in path.get('terminal', []):
                W = _wmat(W, len(ind))
                d = sp.Matrix([xs_sym[i] - _parse(r, {str(th): th}).subs(th, th_s) for i, r in zip(ind, refs)])
                Vp += (d.T * sp.Matrix(W) * d)[0, 0]
            if path.get('u_pf_ref') is not None:
                lp += (uth_s - path['u_pf_ref']) ** 2 * path.get('u_pf_weight', 10.)
            self.x_lb = np.concatenate([self.x_lb, [path.get('theta_lb', 0.)]])
            self.x_ub = np.concatenate([self.x_ub, [path.get('theta_ub', INF)]])
            self.u_lb = np.concatenate([self.u_lb, [path.get('u_pf_lb', 1e-4)]])
            self.u_ub = np.concatenate([self.u_ub, [path.get('u_pf_ub', 1.)]])
            self.x_guess = np.concatenate([self.x_guess, [path.get('theta_guess', 0.)]])
            self.u_guess = np.concatenate([self.u_guess, [path.get('u_pf_lb', 1e-4) + 1e-4]])
        if generic_stage is not None:
            # `nmpc.stage_cost.cost = ...` (modeling.py:38-87): QUIRK restated (oracle/nmpc_coll.py) - the expression is attached after
            # the model was scaled, so its symbols are the SCALED variables
            sub = {s: zs[i] for i, s in enumerate(model.x)}
            sub.update({s: zs[self.nxa + i] for i, s in enumerate(model.u)})
            lp += sp.sympify(generic_stage).subs(sub, simultaneous=True)
        self._lp = self._vgh(lp, zs)
        self._Vp = self._vgh(Vp, xs_sym)
        # ---- constraint rows d(zs, e) with bounds [dlb, dub] ----
        self.ne = 0
        rows, dlb, dub, self.row_ref = [], [], [], []
        es = []
        if constraint:
            sub = {s: zs[i] * self.sza[i] for i, s in enumerate(model.x)}
            sub.update({s: zs[self.nxa + i] * self.su[i] for i, s in enumerate(model.u)})
            cnames = names
            if th is not None:                    # a constraint may involve the path variable: it is a state of the augmented model
                cnames = dict(names, **{str(th): th})
                sub[th] = zs[nx] * self.sza[nx]
            cs = [_parse(e, cnames).subs(sub, simultaneous=True) for e in constraint['expr']]
            nc = len(cs)
            lb = np.broadcast_to(np.asarray(constraint.get('lb', -INF), dtype=float), (nc,))
            ub = np.broadcast_to(np.asarray(constraint.get('ub', INF), dtype=float), (nc,))
            if constraint.get('soft'):
                self.ne = nc
                es = [sp.Symbol(f'e_{j}') for j in range(nc)]
                W = constraint.get('weight')
                self.We = np.diag(np.ones(nc) * 1e4) if W is None else _wmat(W, nc)       # modeling.py:875
                self.e_ub = np.broadcast_to(np.asarray(constraint.get('max_violation', INF), dtype=float), (nc,))
                for j in range(nc):                                                  # mpc.py:1276-1277, :1711-1712
                    for r, (expr, b) in enumerate(((cs[j] - es[j], ub[j]), (-cs[j] - es[j], -lb[j]))):
                        if np.isfinite(b):
                            rows.append(expr), dlb.append(-INF), dub.append(b), self.row_ref.append(r * nc + j)
            else:
                for j in range(nc):
                    if np.isfinite(lb[j]) or np.isfinite(ub[j]):
                        rows.append(cs[j]), dlb.append(lb[j]), dub.append(ub[j]), self.row_ref.append(j)
            self.n_con_ref = 2 * nc if constraint.get('soft') else nc            # rows per stage in the reference's g
        else:
            self.n_con_ref = 0
        self.nrow = len(rows)
        self.dlb, self.dub = np.array(dlb, dtype=float), np.array(dub, dtype=float)
        ze = zs + es
        args = [zs, es]
        self._d = _lam(rows, args) if rows else None
        self._dj = _lam([[sp.diff(r, a) for a in ze] for r in rows], args) if rows else None
        self._dh = _lam([[[sp.diff(r, a, b) for b in zs] for a in zs] for r in rows], args) if rows else None
        # ---- hard terminal constraint on the integrated end state Phi(x_{N-1}, u_{N-1}) (mpc.py:1693-1700), un-scaled ----
        # soft (mpc.py:1684-1692, :1540-1548): rows  c_T(x_{N-1}) - e_T <= ub,  -c_T(x_{N-1}) - e_T <= -lb  on the state the
        # last interval STARTS from, slack e_T in [0, max_violation] behind the stage slack in v, e_T^T W e_T once in J.
        # Terminal row r of the solver: tsign[r] * c_T[texpr[r]](xe) - e_T[texpr[r]] (soft)  in [tlb[r], tub[r]].
        self.nt = self.ne_t = self.n_tcon_ref = 0
        self.t_soft = False
        if terminal_constraint:
            xsym = [sp.Symbol(f'xe_{i}') for i in range(nx)]
            ct = [_parse(e, names).subs(dict(zip(model.x, xsym)), simultaneous=True) for e in terminal_constraint['expr']]
            nct = len(ct)
            lbt = np.broadcast_to(np.asarray(terminal_constraint.get('lb', -INF), dtype=float), (nct,))
            ubt = np.broadcast_to(np.asarray(terminal_constraint.get('ub', INF), dtype=float), (nct,))
            self.t_soft = bool(terminal_constraint.get('soft'))
            self.texpr, self.tsign, self.trow_ref, tlb, tub = [], [], [], [], []
            if self.t_soft:
                self.ne_t = nct
                W = terminal_constraint.get('weight')
                self.WeT = np.diag(np.ones(nct) * 1e4) if W is None else _wmat(W, nct)    # modeling.py:875
                self.eT_ub = np.broadcast_to(np.asarray(terminal_constraint.get('max_violation', INF), dtype=float), (nct,))
                for j in range(nct):
                    for r, (sg, b) in enumerate(((1., ubt[j]), (-1., -lbt[j]))):
                        if np.isfinite(b):
                            self.texpr.append(j), self.tsign.append(sg), tlb.append(-INF), tub.append(b)
                            self.trow_ref.append(r * nct + j)
                self.n_tcon_ref = 2 * nct
            else:
                for j in range(nct):
                    self.texpr.append(j), self.tsign.append(1.), tlb.append(lbt[j]), tub.append(ubt[j])
                    self.trow_ref.append(j)
                self.n_tcon_ref = nct
            self.nt = len(self.texpr)
            self.tsign = np.array(self.tsign)
            self.tlb, self.tub = np.array(tlb, dtype=float), np.array(tub, dtype=float)
            ax = [xsym]
            self._ct = _lam(ct, ax)
            self._ctj = _lam([[sp.diff(c, a) for a in xsym] for c in ct], ax)
            self._cth = _lam([[[sp.diff(c, a, b) for b in xsym] for a in xsym] for c in ct], ax)
        # ---- decision-vector bookkeeping of the reference (mpc.py:1462-1537) ----
        nxa, nua = self.nxa, self.nua
        off = 0
        self.x_ind = []
        for _ in range(N + 1):
            self.x_ind.append(list(range(off, off + nxa)))
            off += nxa
        self.u_ind = []
        for _ in range(N):
            self.u_ind.append(list(range(off, off + nua)))
            off += nua
        self.e_ind = list(range(off, off + self.ne))
        self.eT_ind = list(range(off + self.ne, off + self.ne + self.ne_t))
        self.n_v = off + self.ne + self.ne_t
        self.n_g = N * (nxa + self.n_con_ref) + self.n_tcon_ref
        # ---- custom rows over the whole decision vector ----
        self.n_cus = 0
        self.c_soft = bool(custom and custom.get('soft'))
        if custom:
            vs = [sp.Symbol(f'v_{i}') for i in range(self.n_v)]
            out = custom['fun'](vs, self.x_ind, self.u_ind)
            cs = [sp.sympify(e) for e in (out if isinstance(out, (list, tuple)) else [out])]
            self.n_cus = len(cs)
            self.cus_lb = np.broadcast_to(np.asarray(custom.get('lb', -INF), dtype=float), (self.n_cus,)).copy()
            self.cus_ub = np.broadcast_to(np.asarray(custom.get('ub', INF), dtype=float), (self.n_cus,)).copy()
            if self.c_soft:
                assert not self.ne_t, "soft custom rows next to a soft terminal constraint: not restated"
                n = self.n_cus
                # the slacks take the place of the terminal slacks' block (same treatment: box [0, max_violation], penalty once)
                self.ne_t = n
                self.eT_ind = list(range(self.n_v, self.n_v + n))
                self.n_v += n
                self.WeT = np.diag(np.ones(n) * 1e4)                                            # mpc.py:1732
                self.eT_ub = np.broadcast_to(np.asarray(custom.get('max_violation', INF), dtype=float), (n,)).copy()
                ec = [sp.Symbol(f'v_{i}') for i in self.eT_ind]
                vs = vs + ec
                cs = [c - e for c, e in zip(cs, ec)] + [c + e for c, e in zip(cs, ec)]          # mpc.py:1733-1739
                self.cus_lb, self.cus_ub = (np.concatenate([np.full(n, -INF), self.cus_lb]),
                                            np.concatenate([self.cus_ub, np.full(n, INF)]))
                self.n_cus = 2 * n
            used = sorted({int(str(q)[2:]) for c in cs for q in c.free_symbols})
            self.cus_used = used                                           # entries of v the rows depend on
            us_ = [vs[i] for i in used]
            self._cus = _lam(cs, [us_])
            self._cusj = _lam([[sp.diff(c, a) for a in us_] for c in cs], [us_])
            self._cush = _lam([[[sp.diff(c, a, b) for b in us_] for a in us_] for c in cs], [us_])
            self.n_g += self.n_cus

    @staticmethod
    def _vgh(expr, syms):
        expr = sp.sympify(expr)
        args = [syms]
        if expr == 0:
            n = len(syms)
            return (lambda z: np.zeros(z.shape[0]), lambda z: np.zeros((z.shape[0], n)),
                    lambda z: np.zeros((z.shape[0], n, n)))
        v = _lam([expr], args)
        g = _lam([sp.diff(expr, a) for a in syms], args)
        h = _lam([[sp.diff(expr, a, b) for b in syms] for a in syms], args)
        return (lambda z: v(z)[:, 0], g, h)

    # augmented shooting map on scaled variables: [Phi(x,u); theta + dt u_theta]
    def phia(self, xs, us, p, need=0):
        nx, nu, nth = self.nx, self.nu, self.nth
        B = xs.shape[0]
        if need == 0:
            f = self.phi(xs[:, :nx], us[:, :nu], p)
            if nth:
                f = np.concatenate([f, xs[:, nx:] + self.dt * us[:, nu:]], axis=1)
            return f
        f, J, H = self.phi(xs[:, :nx], us[:, :nu], p, need=2)
        Ja = np.zeros((B, self.nxa, self.nza))
        Ha = np.zeros((B, self.nxa, self.nza, self.nza))
        idx = list(range(nx)) + list(range(self.nxa, self.nxa + nu))
        Ja[np.ix_(range(B), range(nx), idx)] = J
        Ha[np.ix_(range(B), range(nx), idx, idx)] = H
        if nth:
            f = np.concatenate([f, xs[:, nx:] + self.dt * us[:, nu:]], axis=1)
            Ja[:, nx, nx] = 1.0
            Ja[:, nx, self.nxa + nu] = self.dt
        return f, Ja, Ha


class GenIpm(DenseIpm):
    """Free variables w = [x_0 (only with free_x0) | theta_0 | xa_1..xa_N | ua_0..ua_{N-1} | e | s_0..s_{N-1}].
    free_x0: optimize(fix_x0=False) of mpc.py:797-807 - the measured state is not imposed, x_0 lives in the state box."""

    def __init__(self, prob: GenNmpcProblem, options: IpmOptions | None = None, free_x0=False, x0_box=None):
        """x0_box = (lb, ub) in original units: the own box of x_0 of optimize(fix_x0=False, x0_lb=, x0_ub=) (mpc.py:803-807)."""
        self.pb = pb = prob
        self.o = o = options or IpmOptions()
        N, nxa, nua, nth, ne, nrow = pb.N, pb.nxa, pb.nua, pb.nth, pb.ne, pb.nrow
        self.n0 = pb.nx if free_x0 else 0
        self.o_x = self.n0 + nth
        self.o_u = self.o_x + N * nxa
        self.o_e = self.o_u + N * nua
        self.o_eT = self.o_e + ne                            # slack of the soft terminal constraint
        self.o_s = self.o_eT + pb.ne_t
        self.o_t = self.o_s + N * nrow                       # slacks of the terminal rows
        self.o_c = self.o_t + pb.nt                          # slacks of the custom rows (IPOPT's slack form, like every inequality row)
        ncu = getattr(pb, 'n_cus', 0)
        self.nw = self.o_c + ncu
        self.m = N * nxa + N * nrow + pb.nt + ncu
        lb = np.concatenate([pb.x_lb[pb.nx - self.n0:], np.tile(pb.x_lb, N), np.tile(pb.u_lb, N), np.zeros(ne + pb.ne_t), np.tile(pb.dlb, N),
                             pb.tlb if pb.nt else np.zeros(0), pb.cus_lb if ncu else np.zeros(0)])
        ub = np.concatenate([pb.x_ub[pb.nx - self.n0:], np.tile(pb.x_ub, N), np.tile(pb.u_ub, N),
                             pb.e_ub if ne else np.zeros(0), pb.eT_ub if pb.ne_t else np.zeros(0), np.tile(pb.dub, N),
                             pb.tub if pb.nt else np.zeros(0), pb.cus_ub if ncu else np.zeros(0)])
        if x0_box is not None:
            assert free_x0
            lb, ub = lb.copy(), ub.copy()
            lb[:pb.nx] = np.asarray(x0_box[0], dtype=float) / pb.sx
            ub[:pb.nx] = np.asarray(x0_box[1], dtype=float) / pb.sx
        r = o.bound_relax_factor
        self.lb = np.where(np.isfinite(lb), lb - r * np.maximum(1, np.abs(lb)), lb)
        self.ub = np.where(np.isfinite(ub), ub + r * np.maximum(1, np.abs(ub)), ub)
        self.has_l, self.has_u = np.isfinite(self.lb), np.isfinite(self.ub)

    # column of stage k's augmented z entry i in w (-1: pinned x_0 entry)
    def zcols(self, k):
        pb = self.pb
        cols = []
        for i in range(pb.nxa):
            if k == 0:
                cols.append(self.n0 + i - pb.nx if i >= pb.nx else (i if self.n0 else -1))
            else:
                cols.append(self.o_x + (k - 1) * pb.nxa + i)
        cols += [self.o_u + k * pb.nua + j for j in range(pb.nua)]
        return cols

    def _cus_args(self, w, X, U):
        """Values of the entries of v the custom rows read, and the column of each in w (-1: a pinned entry of x_0)."""
        pb = self.pb
        B = w.shape[0]
        v = np.concatenate([X.reshape(B, -1), U.reshape(B, -1), w[:, self.o_e:self.o_s]], axis=1)
        col = np.full(pb.n_v, -1)
        for k in range(pb.N + 1):
            zc = self.zcols(k) if k < pb.N else [self.o_x + (pb.N - 1) * pb.nxa + i for i in range(pb.nxa)]
            for i, j in enumerate(pb.x_ind[k]):
                col[j] = zc[i]
        for k in range(pb.N):
            for i, j in enumerate(pb.u_ind[k]):
                col[j] = self.o_u + k * pb.nua + i
        for a, j in enumerate(pb.e_ind + pb.eT_ind):
            col[j] = self.o_e + a
        return v[:, pb.cus_used], col[pb.cus_used]

    def _unpack(self, w, x0):
        pb = self.pb
        B = w.shape[0]
        N, nxa, nua = pb.N, pb.nxa, pb.nua
        X = np.empty((B, N + 1, nxa))
        X[:, 0, :pb.nx] = w[:, :self.n0] if self.n0 else x0
        X[:, 0, pb.nx:] = w[:, self.n0:self.n0 + pb.nth]
        X[:, 1:] = w[:, self.o_x:self.o_u].reshape(B, N, nxa)
        U = w[:, self.o_u:self.o_e].reshape(B, N, nua)
        E = w[:, self.o_e:self.o_eT]
        S = w[:, self.o_s:self.o_t].reshape(B, N, pb.nrow)
        return X, U, E, S

    def eval_fc(self, w, data):
        pb = self.pb
        x0, p, u_old = data['x0'], data['p'], data.get('u_old')
        X, U, E, S = self._unpack(w, x0)
        B, N = w.shape[0], pb.N
        f = np.zeros(B)
        c = np.empty((B, N, pb.nxa))
        cd = np.empty((B, N, pb.nrow))
        for k in range(N):
            zk = np.concatenate([X[:, k], U[:, k]], axis=1)
            z = zk - pb.zrefa
            f += np.einsum('bi,ij,bj->b', z, pb.Wza, z) + pb._lp[0](zk)
            if pb.ne:
                f += np.einsum('bi,ij,bj->b', E, pb.We, E)                     # mpc.py:1708: once per stage
            c[:, k] = X[:, k + 1] - pb.phia(X[:, k], U[:, k], p)
            if pb.nrow:
                cd[:, k] = pb._d(zk, E) - S[:, k]
        if u_old is not None:
            d = U[:, 0, :pb.nu] - u_old
            f += np.einsum('bi,ij,bj->b', d, pb.Wdu, d)
        d = X[:, N] - pb.xrefNa
        f += np.einsum('bi,ij,bj->b', d, pb.WNa, d) + pb._Vp[0](X[:, N])
        call = np.concatenate([c, cd], axis=2).reshape(B, -1)
        if pb.nt:
            call = np.concatenate([call, self._term_rows(w, X, U, p) - w[:, self.o_t:]], axis=1)
        if pb.ne_t:
            ET = w[:, self.o_eT:self.o_s]
            f += np.einsum('bi,ij,bj->b', ET, pb.WeT, ET)                      # mpc.py:1686: once
        if getattr(pb, 'n_cus', 0):
            va, _ = self._cus_args(w, X, U)
            call = np.concatenate([call, pb._cus(va) - w[:, self.o_c:]], axis=1)
        return f, call

    def _term_rows(self, w, X, U, p):
        """Values of the terminal rows (without their slacks s_T)."""
        pb = self.pb
        if pb.t_soft:
            xe = X[:, pb.N - 1, :pb.nx] * pb.sx
            return pb.tsign * pb._ct(xe)[:, pb.texpr] - w[:, self.o_eT:self.o_s][:, pb.texpr]
        xe = pb.phia(X[:, pb.N - 1], U[:, pb.N - 1], p)[:, :pb.nx] * pb.sx
        return pb._ct(xe)[:, pb.texpr]

    def eval_all(self, w, lam, data):
        """Constraint order: per stage [defect (nxa) | d - s (nrow)]."""
        pb = self.pb
        N, nxa, nua, nza, nrow, ne = pb.N, pb.nxa, pb.nua, pb.nza, pb.nrow, pb.ne
        x0, p, u_old = data['x0'], data['p'], data.get('u_old')
        X, U, E, S = self._unpack(w, x0)
        B = w.shape[0]
        mk = nxa + nrow
        f = np.zeros(B)
        g = np.zeros((B, self.nw))
        c = np.empty((B, N, mk))
        J = np.zeros((B, self.m, self.nw))
        W = np.zeros((B, self.nw, self.nw))
        lam_c = lam[:, N * mk + pb.nt:]                      # custom rows (behind the terminal rows)
        lam_t = lam[:, N * mk:N * mk + pb.nt]
        lam = lam[:, :N * mk].reshape(B, N, mk)
        bi = np.arange(B)
        ecols = list(range(self.o_e, self.o_eT))
        for k in range(N):
            cols = self.zcols(k)
            sel = [i for i, cidx in enumerate(cols) if cidx >= 0]
            zi = [cols[i] for i in sel]
            zk = np.concatenate([X[:, k], U[:, k]], axis=1)
            z = zk - pb.zrefa
            f += np.einsum('bi,ij,bj->b', z, pb.Wza, z) + pb._lp[0](zk)
            gz = 2 * z @ pb.Wza + pb._lp[1](zk)
            Hz = 2 * pb.Wza[None] + pb._lp[2](zk)
            Phi, Jk, Hk = pb.phia(X[:, k], U[:, k], p, need=2)
            c[:, k, :nxa] = X[:, k + 1] - Phi
            Hz = Hz - np.einsum('bm,bmzy->bzy', lam[:, k, :nxa], Hk)
            rows = list(range(k * mk, k * mk + nxa))
            J[np.ix_(bi, rows, zi)] = -Jk[:, :, sel]
            J[:, rows, [self.o_x + k * nxa + i for i in range(nxa)]] = 1.0
            if nrow:
                dv = pb._d(zk, E)
                dj = pb._dj(zk, E)                                              # [B, nrow, nza + ne]
                dh = pb._dh(zk, E)                                              # [B, nrow, nza, nza]
                c[:, k, nxa:] = dv - S[:, k]
                rws = list(range(k * mk + nxa, (k + 1) * mk))
                J[np.ix_(bi, rws, zi)] = dj[:, :, sel]
                if ne:
                    J[np.ix_(bi, rws, ecols)] = dj[:, :, nza:]
                J[:, rws, [self.o_s + k * nrow + r for r in range(nrow)]] = -1.0
                Hz = Hz + np.einsum('bm,bmzy->bzy', lam[:, k, nxa:], dh)
            if ne:
                f += np.einsum('bi,ij,bj->b', E, pb.We, E)
                g[:, ecols] += 2 * E @ pb.We
                W[np.ix_(bi, ecols, ecols)] += 2 * pb.We
            if k == 0 and u_old is not None:
                d = U[:, 0, :pb.nu] - u_old
                f += np.einsum('bi,ij,bj->b', d, pb.Wdu, d)
                gz[:, nxa:nxa + pb.nu] += 2 * d @ pb.Wdu
                Hz[:, nxa:nxa + pb.nu, nxa:nxa + pb.nu] += 2 * pb.Wdu
            if pb.nt and k == N - 1 and pb.t_soft:      # sign c_T(x_{N-1} sx) - e_T - s_T
                nx = pb.nx
                xe = X[:, k, :nx] * pb.sx
                cv, cj, ch = pb._ct(xe), pb._ctj(xe), pb._cth(xe)
                sg = pb.tsign
                ET = w[:, self.o_eT:self.o_s]
                ct_rows = list(range(N * mk, N * mk + pb.nt))
                c_term = sg * cv[:, pb.texpr] - ET[:, pb.texpr] - w[:, self.o_t:]
                Jz = np.zeros((B, pb.nt, nza))
                Jz[:, :, :nx] = sg[None, :, None] * cj[:, pb.texpr] * pb.sx[None, None, :]
                J[np.ix_(bi, ct_rows, zi)] = Jz[:, :, sel]
                J[:, ct_rows, [self.o_eT + j for j in pb.texpr]] = -1.0
                J[:, ct_rows, [self.o_t + r for r in range(pb.nt)]] = -1.0
                Hz[:, :nx, :nx] += np.einsum('bm,bmac->bac', lam_t * sg, ch[:, pb.texpr]) * np.outer(pb.sx, pb.sx)[None]
                tcols = list(range(self.o_eT, self.o_s))
                f += np.einsum('bi,ij,bj->b', ET, pb.WeT, ET)
                g[:, tcols] += 2 * ET @ pb.WeT
                W[np.ix_(bi, tcols, tcols)] += 2 * pb.WeT
            elif pb.nt and k == N - 1:      # c_T(Phi(z) sx) - s_T: chain rule through the shooting map
                nx = pb.nx
                xe = Phi[:, :nx] * pb.sx
                cv, cj, ch = pb._ct(xe), pb._ctj(xe), pb._cth(xe)
                Js = Jk[:, :nx] * pb.sx[None, :, None]                                   # d xe / d z
                ct_rows = list(range(N * mk, N * mk + pb.nt))
                c_term = cv - w[:, self.o_t:]
                J[np.ix_(bi, ct_rows, zi)] = np.einsum('bma,baz->bmz', cj, Js)[:, :, sel]
                J[:, ct_rows, [self.o_t + r for r in range(pb.nt)]] = -1.0
                Hs = Hk[:, :nx] * pb.sx[None, :, None, None]
                Hz = Hz + np.einsum('bm,bmac,baz,bcy->bzy', lam_t, ch, Js, Js) + \
                    np.einsum('bm,bma,bazy->bzy', lam_t, cj, Hs)
            g[:, zi] += gz[:, sel]
            W[np.ix_(bi, zi, zi)] += Hz[np.ix_(bi, sel, sel)]
        if pb.ne_t and not pb.t_soft:       # slacks of soft custom rows: 1e4 e^T e once (mpc.py:1732-1733)
            ET = w[:, self.o_eT:self.o_s]
            tcols = list(range(self.o_eT, self.o_s))
            f += np.einsum('bi,ij,bj->b', ET, pb.WeT, ET)
            g[:, tcols] += 2 * ET @ pb.WeT
            W[np.ix_(bi, tcols, tcols)] += 2 * pb.WeT
        d = X[:, N] - pb.xrefNa
        xi = [self.o_x + (N - 1) * nxa + i for i in range(nxa)]
        f += np.einsum('bi,ij,bj->b', d, pb.WNa, d) + pb._Vp[0](X[:, N])
        g[:, xi] += 2 * d @ pb.WNa + pb._Vp[1](X[:, N])
        W[np.ix_(bi, xi, xi)] += 2 * pb.WNa[None] + pb._Vp[2](X[:, N])
        call = c.reshape(B, -1)
        if pb.nt:
            call = np.concatenate([call, c_term], axis=1)
        ncu = getattr(pb, 'n_cus', 0)
        if ncu:
            va, col = self._cus_args(w, X, U)
            keep = [q for q, cc in enumerate(col) if cc >= 0]
            wc = [int(col[q]) for q in keep]
            rws = list(range(self.m - ncu, self.m))
            call = np.concatenate([call, pb._cus(va) - w[:, self.o_c:]], axis=1)
            J[np.ix_(bi, rws, wc)] = pb._cusj(va)[:, :, keep]
            J[:, rws, [self.o_c + r for r in range(ncu)]] = -1.0
            W[np.ix_(bi, wc, wc)] += np.einsum('bm,bmac->bac', lam_c, pb._cush(va))[np.ix_(bi, keep, keep)]
        return f, g, call, J, W

    def solve(self, x0, p, w0=None, u_old=None, verbose=False):
        """w0: warm start for [theta_0 | xa | ua | e] (the slacks always restart at d(w_0), like IPOPT)."""
        o, pb = self.o, self.pb
        x0 = np.atleast_2d(np.asarray(x0, dtype=float)) / pb.sx
        B = x0.shape[0]
        p = np.broadcast_to(np.atleast_2d(np.asarray(p, dtype=float)), (B, pb.np_)) if pb.np_ else np.zeros((B, 0))
        data = {'x0': x0, 'p': p}
        if u_old is not None:
            data['u_old'] = np.broadcast_to(np.atleast_2d(np.asarray(u_old, dtype=float)), (B, pb.nu))
        if w0 is None:
            w0 = np.concatenate([pb.x_guess[pb.nx - self.n0:], np.tile(pb.x_guess, pb.N), np.tile(pb.u_guess, pb.N), np.zeros(pb.ne),
                                 np.full(pb.ne_t, float(pb.ne_t) if getattr(pb, 'c_soft', False) else 0.)])           # mpc.py:1555
        w0 = np.broadcast_to(np.atleast_2d(w0)[:, :self.o_s], (B, self.o_s))
        w0 = _push_interior(w0, self.lb[:self.o_s], self.ub[:self.o_s], o)
        if pb.nrow or pb.nt:
            X, U, E, _ = self._unpack(np.concatenate([w0, np.zeros((B, pb.N * pb.nrow + pb.nt))], axis=1), x0)
            if pb.nrow:
                s0 = np.stack([pb._d(np.concatenate([X[:, k], U[:, k]], axis=1), E) for k in range(pb.N)], axis=1)
                w0 = np.concatenate([w0, s0.reshape(B, -1)], axis=1)
            if pb.nt:
                wz = np.concatenate([w0, np.zeros((B, self.nw - w0.shape[1]))], axis=1)
                w0 = np.concatenate([w0, self._term_rows(wz, X, U, p)], axis=1)
        if getattr(pb, 'n_cus', 0):      # slacks of the custom rows start at the rows' values, like every slack
            wz = np.concatenate([w0, np.zeros((B, self.nw - w0.shape[1]))], axis=1)
            X, U, _, _ = self._unpack(wz, x0)
            w0 = np.concatenate([w0, pb._cus(self._cus_args(wz, X, U)[0])], axis=1)
        res = self.solve_data(data, w0, verbose)
        X, U, E, S = self._unpack(res['w'], x0)
        res.update(X=X, U=U, E=E, S=S, u0=U[:, 0, :pb.nu] * pb.su, x0=x0)
        return res

    # reference layouts -------------------------------------------------------------------------------------------
    def to_v(self, res):
        B = res['X'].shape[0]
        return np.concatenate([res['X'].reshape(B, -1), res['U'].reshape(B, -1), res['E'], res['w'][:, self.o_eT:self.o_s]], axis=1)

    def w_from_v(self, v):
        """[theta_0 | xa_1.. | ua | e] from the reference's decision vector."""
        pb = self.pb
        v = np.atleast_2d(v)
        nX = (pb.N + 1) * pb.nxa
        return np.concatenate([v[:, pb.nx - self.n0:pb.nxa], v[:, pb.nxa:nX], v[:, nX:]], axis=1)

    def lam_g(self, res):
        """Multipliers in the reference's g order: per stage [defect (nxa) | constraint rows (n_con_ref)]; rows that were
        dropped (infinite bound) carry a zero multiplier."""
        pb = self.pb
        B = res['lam'].shape[0]
        mk = pb.nxa + pb.nrow
        lam = res['lam'][:, :pb.N * mk].reshape(B, pb.N, mk)
        out = np.zeros((B, pb.N, pb.nxa + pb.n_con_ref))
        out[:, :, :pb.nxa] = lam[:, :, :pb.nxa]
        for r, ref in enumerate(pb.row_ref):
            out[:, :, pb.nxa + ref] = lam[:, :, pb.nxa + r]
        ncu = getattr(pb, 'n_cus', 0)
        lam_cus = res['lam'][:, pb.N * mk + pb.nt:pb.N * mk + pb.nt + ncu]  # custom rows: the last rows of g (mpc.py:1741-1744)
        if not pb.nt:
            return np.concatenate([out.reshape(B, -1), lam_cus], axis=1)
        # last stage: [defect | terminal rows | stage rows] (mpc.py:1693-1700 before :1707)
        head = out[:, :-1].reshape(B, -1)
        last = out[:, -1]
        lt = np.zeros((B, pb.n_tcon_ref))                                   # dropped (unbounded) rows: zero multiplier
        lt[:, pb.trow_ref] = res['lam'][:, pb.N * mk:pb.N * mk + pb.nt]
        return np.concatenate([head, last[:, :pb.nxa], lt, last[:, pb.nxa:], lam_cus], axis=1)
