"""Oracle: direct multiple-shooting NMPC - transcription + dense primal-dual interior-point solver, numpy.

TEST INFRASTRUCTURE ONLY - never imported by the product package.

PARITY: pinned for the interior-point method and the collocation transcription by the reference's CSTR notebook (see
oracle/nmpc_coll.py); this module's multiple-shooting transcription itself has no reference number to hold on to - PARITY
UNPINNED for it: the reference's NMPC tests hold no numeric assertion (tests/test_NMPC.py are closed-loop smoke
tests) and its solver, IPOPT, lives in the un-vendored, un-installable dependency `casadi>=3.5` (setup.py:52).
This file therefore restates
  (1) the reference's *transcription* for a pre-discretised model with `integration_method='discrete'`
      (hilo_mpc/modules/controller/mpc.py:1455-1787; recipe of SURVEY Q18): decision vector
      v = [x_0..x_N | u_0..u_{Nc-1}] in scaled variables (:1462-1485), equality rows x_{k+1} - Phi(x_k,u_k) (:1667),
      objective sum_k l(x_k,u_k) + V(Phi_{N-1}) (:1676-1682), x_0 pinned through its bounds (:797-802), quadratic
      costs of `QuadraticCost` (hilo_mpc/util/modeling.py:243-283: (s-r)^T W (s-r), no factor 1/2; references divided
      by the scaling :310; the input-change term only acts in interval 0, mpc.py:1631-1635), scaling of
      bounds/guesses (mpc.py:248-263) and of the model (hilo_mpc/modules/base.py:1562-1591), and
  (2) the published interior-point algorithm IPOPT implements (Waechter & Biegler, Math. Program. 106, 2006):
      monotone barrier update (their eq. 7), fraction-to-the-boundary rule (8), filter line search (Alg. A,
      without second-order correction / restoration phase), inertia correction (Alg. IC), scaled optimality
      error E_mu (5,6), default constants (tol 1e-8, mu_0 0.1, kappa_eps 10, kappa_mu 0.2, theta_mu 1.5,
      tau_min 0.99, bound_push = bound_frac 1e-2, bound_relax_factor 1e-8, max_iter 3000).
Its results are cross-checked in tests/ by an independent solver (scipy SLSQP / trust-constr) on the same NLP and
by the KKT residual at the returned point.
"""
from __future__ import annotations

import numpy as np

from .shooting import ShootingMap

INF = np.inf

# solver status codes, hilo_mpc/modules/optimizer.py:1085-1104
SOLVED, ACCEPTABLE, INFEASIBLE, RESTORATION_FAILED, MAXITER, OTHER = 1, 2, 3, 4, 5, -1


def _wmat(W, n):
    """QuadraticCost._create_weight_matrix (modeling.py:164-185)."""
    W = np.asarray(W, dtype=float)
    if W.ndim == 0:
        W = np.diag([float(W)] * n) if n > 1 else np.array([[float(W)]])
    elif W.ndim == 1:
        W = np.diag(W)
    assert W.shape == (n, n)
    return W


class NmpcProblem:
    """Plain description of one NMPC structure (everything `NMPC.setup()` fixes)."""

    def __init__(self, model, dt, N, order=4, n_sub=1, Nc=None,
                 stage_states=None, stage_inputs=None, input_change=None, terminal_states=None,
                 x_lb=None, x_ub=None, u_lb=None, u_ub=None, x_scaling=None, u_scaling=None,
                 x_guess=None, u_guess=None):
        self.model, self.dt, self.N, self.order = model, float(dt), int(N), order
        self.Nc = self.N if Nc is None else int(Nc)
        nx, nu = model.nx, model.nu
        self.nx, self.nu, self.np_ = nx, nu, model.np_
        self.nz = nx + nu
        self.sx = np.ones(nx) if x_scaling is None else np.asarray(x_scaling, dtype=float)
        self.su = np.ones(nu) if u_scaling is None else np.asarray(u_scaling, dtype=float)
        # mpc.py:645-701 defaults +-inf; :248-263 bounds and guesses are divided by the scaling
        self.x_lb = (np.full(nx, -INF) if x_lb is None else np.asarray(x_lb, dtype=float)) / self.sx
        self.x_ub = (np.full(nx, INF) if x_ub is None else np.asarray(x_ub, dtype=float)) / self.sx
        self.u_lb = (np.full(nu, -INF) if u_lb is None else np.asarray(u_lb, dtype=float)) / self.su
        self.u_ub = (np.full(nu, INF) if u_ub is None else np.asarray(u_ub, dtype=float)) / self.su
        self.x_guess = (np.zeros(nx) if x_guess is None else np.asarray(x_guess, dtype=float)) / self.sx
        self.u_guess = (np.zeros(nu) if u_guess is None else np.asarray(u_guess, dtype=float)) / self.su
        self.smap = ShootingMap(model, order, n_sub)

        # ---- quadratic cost as (z - zref)^T Wz (z - zref) on scaled z = (x, u); refs are divided by the scaling
        self.Wz = np.zeros((self.nz, self.nz))
        self.zref = np.zeros(self.nz)
        for ind, W, ref in (stage_states or []):
            ind = list(ind)
            self.Wz[np.ix_(ind, ind)] += _wmat(W, len(ind))
            if ref is not None:
                self.zref[ind] = np.asarray(ref, dtype=float) / self.sx[ind]       # modeling.py:310
        for ind, W, ref in (stage_inputs or []):
            ind = list(ind)
            jj = [nx + i for i in ind]
            self.Wz[np.ix_(jj, jj)] += _wmat(W, len(ind))
            if ref is not None:
                self.zref[jj] = np.asarray(ref, dtype=float) / self.su[ind]
        self.Wdu = np.zeros((nu, nu))
        for ind, W in ([input_change] if input_change else []):
            ind = list(ind)
            self.Wdu[np.ix_(ind, ind)] += _wmat(W, len(ind))
        self.WN = np.zeros((nx, nx))
        self.xrefN = np.zeros(nx)
        for ind, W, ref in (terminal_states or []):
            ind = list(ind)
            self.WN[np.ix_(ind, ind)] += _wmat(W, len(ind))
            if ref is not None:
                self.xrefN[ind] = np.asarray(ref, dtype=float) / self.sx[ind]

        # ---- decision-vector bookkeeping, bit-exact restatement of mpc.py:1462-1485 (integer index maps) ----
        N, Nc = self.N, self.Nc
        off = 0
        self.x_ind = []
        for _ in range(N + 1):
            self.x_ind.append(list(range(off, off + nx)))
            off += nx
        self.u_ind = []
        for _ in range(Nc):
            self.u_ind.append(list(range(off, off + nu)))
            off += nu
        self.n_v = off                                           # mpc.py:1440
        self.n_g = N * nx                                        # mpc.py:1667-1669
        self.v_lb = np.concatenate([np.tile(self.x_lb, N + 1), np.tile(self.u_lb, Nc)])
        self.v_ub = np.concatenate([np.tile(self.x_ub, N + 1), np.tile(self.u_ub, Nc)])
        self.v_guess = np.concatenate([np.tile(self.x_guess, N + 1), np.tile(self.u_guess, Nc)])

    # ---- scaled shooting map (base.py:1562-1591: x := x*s inside the equations, rhs := rhs/s) ----------------
    def phi(self, xs, us, p, need=0):
        x = xs * self.sx
        u = us * self.su
        if need == 0:
            return self.smap.value(x, u, p, self.dt) / self.sx
        f, J, H = self.smap(x, u, p, self.dt)
        sz = np.concatenate([self.sx, self.su])
        J = J * sz[None, None, :] / self.sx[None, :, None]
        H = H * sz[None, None, :, None] * sz[None, None, None, :] / self.sx[None, :, None, None]
        return f / self.sx, J, H

    def u_of(self, U, k):
        """mpc.py:1629-1630: beyond the control horizon the last input is held."""
        return U[:, min(k, self.Nc - 1)]

    # ---- reference-layout NLP functions on v (for the independent scipy cross-check) ------------------------
    def split(self, v):
        v = np.atleast_2d(v)
        X = v[:, :(self.N + 1) * self.nx].reshape(-1, self.N + 1, self.nx)
        U = v[:, (self.N + 1) * self.nx:].reshape(-1, self.Nc, self.nu)
        return X, U

    def join(self, X, U):
        return np.concatenate([X.reshape(X.shape[0], -1), U.reshape(U.shape[0], -1)], axis=1)

    def objective(self, v, p, u_old=None):
        """mpc.py:1676-1682: J = sum_k l(x_k,u_k) + V(Phi(x_{N-1},u_{N-1}))."""
        X, U = self.split(v)
        B = X.shape[0]
        J = np.zeros(B)
        for k in range(self.N):
            z = np.concatenate([X[:, k], self.u_of(U, k)], axis=1) - self.zref
            J += np.einsum('bi,ij,bj->b', z, self.Wz, z)
            if k == 0 and u_old is not None:                     # mpc.py:1631-1635
                d = U[:, 0] - np.atleast_2d(u_old)
                J += np.einsum('bi,ij,bj->b', d, self.Wdu, d)
        xN = self.phi(X[:, self.N - 1], self.u_of(U, self.N - 1), p)
        d = xN - self.xrefN
        return J + np.einsum('bi,ij,bj->b', d, self.WN, d)

    def constraints(self, v, p):
        """mpc.py:1667: g = [x_{k+1} - Phi(x_k,u_k)]_k."""
        X, U = self.split(v)
        g = [X[:, k + 1] - self.phi(X[:, k], self.u_of(U, k), p) for k in range(self.N)]
        return np.concatenate(g, axis=1)


# ==================================================================================================
# dense primal-dual interior point (batched over instances)
# ==================================================================================================
class IpmOptions:
    tol = 1e-8
    acceptable_tol = 1e-6
    acceptable_iter = 15
    max_iter = 3000
    mu_init = 0.1
    kappa_eps = 10.
    kappa_mu = 0.2
    theta_mu = 1.5
    tau_min = 0.99
    bound_push = 1e-2
    bound_frac = 1e-2
    bound_relax_factor = 1e-8
    s_max = 100.
    kappa_sigma = 1e10
    # filter line search
    gamma_theta = 1e-5
    gamma_phi = 1e-8
    delta = 1.
    s_theta = 1.1
    s_phi = 2.3
    eta_phi = 1e-8
    theta_min_fact = 1e-4
    theta_max_fact = 1e4
    alpha_red = 0.5
    alpha_min_frac = 0.05
    max_filter = 16
    # inertia correction
    delta_w_min = 1e-20
    delta_w_0 = 1e-4
    delta_w_max = 1e40
    kappa_w_minus = 1. / 3
    kappa_w_plus = 8.
    kappa_w_plus_bar = 100.

    compl_inf_tol = 1e-4
    constr_mult_init_max = 1e3      # IPOPT: least-squares estimate of the equality multipliers at the start, dropped if larger
    max_soc = 4                     # second-order correction steps (W&B sec. 2.4), kappa_soc = 0.99
    kappa_soc = 0.99
    ls_mult_init = False            # option; the device engine starts from lambda = 0, so does the oracle by default
    soc = True
    resto_barrier = True

    def __init__(self, **kw):
        for k, v in kw.items():
            if not hasattr(self, k):
                raise ValueError(f"unknown option {k}")
            setattr(self, k, v)

    def mu_floor(self):
        """Smallest barrier parameter of the monotone update.  W&B eq. (7) writes eps_tol / 10; IPOPT's implementation
        (MonotoneMuUpdate::CalcNewMuAndTau) uses min(tol, compl_inf_tol) / (barrier_tol_factor + 1) = tol / 11 with the
        defaults - the value that reproduces the last printed digit of the CSTR notebook's input (59882.1817)."""
        return min(self.tol, self.compl_inf_tol) / (self.kappa_eps + 1.)


def _inertia_ok(K, n_pos, n_neg):
    """Inertia of the symmetric KKT matrix from a Bunch-Kaufman LDL^T factorisation (what IPOPT reads off its
    symmetric indefinite solver): 1x1 pivots by sign, 2x2 pivots by their eigenvalue signs.  An eigenvalue
    decomposition of the whole matrix loses the small eigenvalues next to barrier terms of 1e9 and more."""
    from scipy.linalg import ldl
    out = np.zeros(K.shape[0], dtype=bool)
    for b in range(K.shape[0]):
        try:
            _, D, _ = ldl(K[b], lower=True)
        except Exception:
            continue
        n = D.shape[0]
        pos = neg = 0
        i = 0
        while i < n:
            if i + 1 < n and D[i + 1, i] != 0.0:
                a, c, d = D[i, i], D[i + 1, i], D[i + 1, i + 1]
                det, tr = a * d - c * c, a + d
                if det < 0:
                    pos += 1
                    neg += 1
                elif det > 0:
                    pos += 2 * (tr > 0)
                    neg += 2 * (tr < 0)
                i += 2
            else:
                pos += D[i, i] > 0
                neg += D[i, i] < 0
                i += 1
        out[b] = (pos == n_pos) and (neg == n_neg)
    return out


def _push_interior(w, lb, ub, o):
    """IPOPT initialisation (Waechter & Biegler sec. 3.6): x <- P[x] with kappa_1 = kappa_2 = bound_push/frac."""
    w = w.copy()
    has_l, has_u = np.isfinite(lb), np.isfinite(ub)
    both = has_l & has_u
    pl = np.where(both, np.minimum(o.bound_push * np.maximum(1, np.abs(lb)), o.bound_frac * (ub - lb)),
                  o.bound_push * np.maximum(1, np.abs(lb)))
    pu = np.where(both, np.minimum(o.bound_push * np.maximum(1, np.abs(ub)), o.bound_frac * (ub - lb)),
                  o.bound_push * np.maximum(1, np.abs(ub)))
    with np.errstate(invalid='ignore'):
        w = np.where(has_l, np.maximum(w, lb + pl), w)
        w = np.where(has_u, np.minimum(w, ub - pu), w)
    return w


class DenseIpm:
    """Solves min f(w) s.t. c(w) = 0, l <= w <= u for a batch of NMPC instances.

    Free variables w = [x_1..x_N | u_0..u_{N-1}] (x_0 is fixed by its bounds, mpc.py:797-802, and removed like
    IPOPT's default `fixed_variable_treatment = make_parameter`).  A control horizon Nc < N holds Nc input blocks; stage
    k >= Nc uses the last one (mpc.py:1629-1630)."""

    def __init__(self, prob: NmpcProblem, options: IpmOptions | None = None):
        self.pb = prob
        self.o = options or IpmOptions()
        N, nx, nu, Nc = prob.N, prob.nx, prob.nu, prob.Nc
        self.nw = N * nx + Nc * nu
        self.m = N * nx
        self.ix = [list(range(k * nx, (k + 1) * nx)) for k in range(N)]            # x_{k+1}
        self.iu = [list(range(N * nx + min(k, Nc - 1) * nu, N * nx + (min(k, Nc - 1) + 1) * nu)) for k in range(N)]
        lb = np.concatenate([np.tile(prob.x_lb, N), np.tile(prob.u_lb, Nc)])
        ub = np.concatenate([np.tile(prob.x_ub, N), np.tile(prob.u_ub, Nc)])
        r = self.o.bound_relax_factor
        self.lb = np.where(np.isfinite(lb), lb - r * np.maximum(1, np.abs(lb)), lb)
        self.ub = np.where(np.isfinite(ub), ub + r * np.maximum(1, np.abs(ub)), ub)
        self.has_l, self.has_u = np.isfinite(self.lb), np.isfinite(self.ub)

    # ---- stage-wise evaluation -------------------------------------------------------------------------------
    def _XU(self, w, x0):
        pb = self.pb
        B = w.shape[0]
        X = np.concatenate([x0[:, None, :], w[:, :pb.N * pb.nx].reshape(B, pb.N, pb.nx)], axis=1)
        Uc = w[:, pb.N * pb.nx:].reshape(B, pb.Nc, pb.nu)
        U = Uc[:, np.minimum(np.arange(pb.N), pb.Nc - 1)]      # per stage: the held input beyond the control horizon
        return X, U

    def eval_fc(self, w, data):
        """objective (with the terminal term on x_N, which equals Phi_{N-1} on the feasible set) and defects."""
        pb = self.pb
        x0, p, u_old = data['x0'], data['p'], data.get('u_old')
        # trajectory tracking / time-varying parameters (mpc.py:365-463, :335-364): per-stage references zref_k [B,N,nz]
        # (already divided by the scaling), terminal reference [B,nx], per-stage parameters p_k [B,N,np]
        zr, xrN, pk = data.get('zref_k'), data.get('xrefN'), data.get('p_k')
        X, U = self._XU(w, x0)
        B = w.shape[0]
        f = np.zeros(B)
        c = np.empty((B, pb.N, pb.nx))
        for k in range(pb.N):
            z = np.concatenate([X[:, k], U[:, k]], axis=1) - (pb.zref if zr is None else zr[:, k])
            f += np.einsum('bi,ij,bj->b', z, pb.Wz, z)
            c[:, k] = X[:, k + 1] - pb.phi(X[:, k], U[:, k], p if pk is None else pk[:, k])
        if u_old is not None:
            d = U[:, 0] - u_old
            f += np.einsum('bi,ij,bj->b', d, pb.Wdu, d)
        d = X[:, pb.N] - (pb.xrefN if xrN is None else xrN)
        f += np.einsum('bi,ij,bj->b', d, pb.WN, d)
        return f, c.reshape(B, -1)

    def eval_all(self, w, lam, data):
        """f, grad f, c, J (dense), W = hess_ww (f + lam^T c) (dense)."""
        pb = self.pb
        N, nx, nu, nz = pb.N, pb.nx, pb.nu, pb.nz
        x0, p, u_old = data['x0'], data['p'], data.get('u_old')
        zr, xrN, pk = data.get('zref_k'), data.get('xrefN'), data.get('p_k')
        X, U = self._XU(w, x0)
        B = w.shape[0]
        f = np.zeros(B)
        g = np.zeros((B, self.nw))
        c = np.empty((B, N, nx))
        J = np.zeros((B, self.m, self.nw))
        W = np.zeros((B, self.nw, self.nw))
        lam = lam.reshape(B, N, nx)
        for k in range(N):
            zi = (self.ix[k - 1] if k > 0 else []) + self.iu[k]           # free columns of stage k's z
            sel = (list(range(nx)) if k > 0 else []) + list(range(nx, nz))
            z = np.concatenate([X[:, k], U[:, k]], axis=1) - (pb.zref if zr is None else zr[:, k])
            f += np.einsum('bi,ij,bj->b', z, pb.Wz, z)
            gz = 2 * z @ pb.Wz
            Hz = np.broadcast_to(2 * pb.Wz, (B, nz, nz)).copy()
            Phi, Jk, Hk = pb.phi(X[:, k], U[:, k], p if pk is None else pk[:, k], need=2)
            c[:, k] = X[:, k + 1] - Phi
            Hz -= np.einsum('bm,bmzy->bzy', lam[:, k], Hk)
            if k == 0 and u_old is not None:
                d = U[:, 0] - u_old
                f += np.einsum('bi,ij,bj->b', d, pb.Wdu, d)
                gz[:, nx:] += 2 * d @ pb.Wdu
                Hz[:, nx:, nx:] += 2 * pb.Wdu
            g[:, zi] += gz[:, sel]
            W[np.ix_(range(B), zi, zi)] += Hz[np.ix_(range(B), sel, sel)]
            rows = list(range(k * nx, (k + 1) * nx))
            J[np.ix_(range(B), rows, zi)] += -Jk[:, :, sel]
            J[:, rows, self.ix[k]] = 1.0
        d = X[:, N] - (pb.xrefN if xrN is None else xrN)
        f += np.einsum('bi,ij,bj->b', d, pb.WN, d)
        g[:, self.ix[N - 1]] += 2 * d @ pb.WN
        W[np.ix_(range(B), self.ix[N - 1], self.ix[N - 1])] += 2 * pb.WN
        return f, g, c.reshape(B, -1), J, W

    # ---- barrier pieces ---------------------------------------------------------------------------------------
    def _slacks(self, w):
        sl = np.where(self.has_l, w - self.lb, 1.0)
        su = np.where(self.has_u, self.ub - w, 1.0)
        return sl, su

    def barrier(self, f, w, mu):
        sl, su = self._slacks(w)
        with np.errstate(invalid='ignore', divide='ignore'):
            return f - mu * (np.where(self.has_l, np.log(sl), 0).sum(1) + np.where(self.has_u, np.log(su), 0).sum(1))

    def errors(self, g, c, J, lam, zl, zu, w, mu):
        o = self.o
        sl, su = self._slacks(w)
        r_d = g + np.einsum('bmi,bm->bi', J, lam) - zl + zu
        dual = np.abs(r_d).max(1)
        prim = np.abs(c).max(1) if c.shape[1] else np.zeros(w.shape[0])
        cl = np.where(self.has_l, np.abs(sl * zl - mu[:, None]), 0).max(1)
        cu = np.where(self.has_u, np.abs(su * zu - mu[:, None]), 0).max(1)
        nb = max(1, int(self.has_l.sum() + self.has_u.sum()))
        zsum = np.abs(zl).sum(1) + np.abs(zu).sum(1)
        s_d = np.maximum(o.s_max, (np.abs(lam).sum(1) + zsum) / (self.m + nb)) / o.s_max
        s_c = np.maximum(o.s_max, zsum / nb) / o.s_max
        return np.maximum.reduce([dual / s_d, prim, np.maximum(cl, cu) / s_c]), dual, prim, np.maximum(cl, cu)

    # ---- main loop --------------------------------------------------------------------------------------------
    def solve(self, x0, p, w0=None, u_old=None, verbose=False, zref_k=None, xrefN=None, p_k=None):
        """x0 [B,nx] original units; p [B,np] or [np]; w0 optional warm start (free variables, scaled).
        zref_k [N,nz] / xrefN [nx]: per-stage / terminal references in ORIGINAL units (divided by the scaling here like
        modeling.py:329); p_k [N,np]: per-stage model parameters.  Returns dict(w, lam, zl, zu, f, status, iters, X, U, kkt)."""
        o, pb = self.o, self.pb
        x0 = np.atleast_2d(np.asarray(x0, dtype=float)) / pb.sx
        B = x0.shape[0]
        p = np.broadcast_to(np.atleast_2d(np.asarray(p, dtype=float)), (B, pb.np_)) if pb.np_ else np.zeros((B, 0))
        if u_old is not None:
            u_old = np.broadcast_to(np.atleast_2d(np.asarray(u_old, dtype=float)), (B, pb.nu))
        if w0 is None:
            w0 = np.concatenate([np.tile(pb.x_guess, pb.N), np.tile(pb.u_guess, pb.Nc)])
        data = {'x0': x0, 'p': p}
        if u_old is not None:
            data['u_old'] = u_old
        sz = np.concatenate([pb.sx, pb.su])
        if zref_k is not None:
            data['zref_k'] = np.broadcast_to(np.asarray(zref_k, dtype=float) / sz, (B, pb.N, pb.nz))
        if xrefN is not None:
            data['xrefN'] = np.broadcast_to(np.asarray(xrefN, dtype=float) / pb.sx, (B, pb.nx))
        if p_k is not None:
            data['p_k'] = np.broadcast_to(np.asarray(p_k, dtype=float), (B, pb.N, pb.np_))
        res = self.solve_data(data, w0, verbose)
        X, U = self._XU(res['w'], x0)
        res.update(X=X, U=U, u0=U[:, 0] * pb.su)
        return res

    @staticmethod
    def _sl(data, idx):
        return {k: v[idx] for k, v in data.items()}

    def solve_data(self, data, w0, verbose=False):
        """Generic driver: `data` is a dict of per-instance arrays handed to eval_fc / eval_all."""
        o = self.o
        B = next(iter(data.values())).shape[0]
        w = _push_interior(np.broadcast_to(np.atleast_2d(w0), (B, self.nw)), self.lb, self.ub, o)
        lam = np.zeros((B, self.m))
        zl = np.where(self.has_l, 1.0, 0.0) * np.ones((B, 1))
        zu = np.where(self.has_u, 1.0, 0.0) * np.ones((B, 1))
        mu = np.full(B, o.mu_init)
        tau = np.maximum(o.tau_min, 1 - mu)
        status = np.zeros(B, dtype=np.int32)
        iters = np.zeros(B, dtype=np.int32)
        acc_count = np.zeros(B, dtype=np.int32)
        delta_last = np.zeros(B)
        n_resto = np.zeros(B, dtype=np.int32)
        soc_used = np.zeros(B, dtype=np.int32)
        filt = [[] for _ in range(B)]
        if o.ls_mult_init and self.m:
            # W&B sec. 3.6: lambda_0 from  [I J^T; J 0] [w; lambda] = -[grad f - zL + zU; 0]; discarded when it is large
            _, g0, _, J0, _ = self.eval_all(w, lam, data)
            K = np.zeros((B, self.nw + self.m, self.nw + self.m))
            K[:, :self.nw, :self.nw] = np.eye(self.nw)
            K[:, :self.nw, self.nw:] = np.swapaxes(J0, 1, 2)
            K[:, self.nw:, :self.nw] = J0
            rhs = np.concatenate([-(g0 - zl + zu), np.zeros((B, self.m))], axis=1)
            try:
                lam0 = np.linalg.solve(K, rhs[:, :, None])[:, self.nw:, 0]
                ok = np.isfinite(lam0).all(1) & (np.abs(lam0).max(1) <= o.constr_mult_init_max)
                lam = np.where(ok[:, None], lam0, 0.0)
            except np.linalg.LinAlgError:
                pass
        f, c = self.eval_fc(w, data)
        theta0 = np.abs(c).sum(1)
        theta_min = o.theta_min_fact * np.maximum(1, theta0)
        theta_max = o.theta_max_fact * np.maximum(1, theta0)
        active = np.ones(B, dtype=bool)

        for it in range(o.max_iter + 1):
            idx = np.nonzero(active)[0]
            if idx.size == 0:
                break
            f_a, g, c, J, Wl = self.eval_all(w[idx], lam[idx], self._sl(data, idx))
            E0, dual, prim, compl = self.errors(g, c, J, lam[idx], zl[idx], zu[idx], w[idx], np.zeros(idx.size))
            done = E0 <= o.tol
            status[idx[done]] = SOLVED
            acc = (E0 <= o.acceptable_tol) & ~done
            acc_count[idx] = np.where(acc, acc_count[idx] + 1, 0)
            acc_done = acc_count[idx] >= o.acceptable_iter
            status[idx[acc_done & ~done]] = ACCEPTABLE
            fin = done | acc_done
            if it == o.max_iter:
                status[idx[~fin]] = MAXITER
                fin = np.ones_like(fin)
            active[idx[fin]] = False
            if verbose:
                print(f"it {it:3d} active {idx.size:4d} E0 max {E0.max():.3e} mu {mu[idx].max():.1e}")
            keep = ~fin
            if not keep.any():
                continue
            idx, f_a, g, c, J, Wl = idx[keep], f_a[keep], g[keep], c[keep], J[keep], Wl[keep]
            nb = idx.size
            # ---- barrier parameter update (monotone; W&B eq. 7) ----
            for _ in range(20):
                Emu = self.errors(g, c, J, lam[idx], zl[idx], zu[idx], w[idx], mu[idx])[0]
                dec = (Emu <= o.kappa_eps * mu[idx]) & (mu[idx] > o.mu_floor() * (1 + 1e-12))
                if not dec.any():
                    break
                j = idx[dec]
                mu[j] = np.maximum(o.mu_floor(), np.minimum(o.kappa_mu * mu[j], mu[j] ** o.theta_mu))
                tau[j] = np.maximum(o.tau_min, 1 - mu[j])
                for b in j:
                    filt[b] = []
            # ---- search direction with inertia correction ----
            sl, su = self._slacks(w[idx])
            Sig = np.where(self.has_l, zl[idx] / sl, 0) + np.where(self.has_u, zu[idx] / su, 0)
            rhs1 = -(g - np.where(self.has_l, mu[idx, None] / sl, 0) + np.where(self.has_u, mu[idx, None] / su, 0))
            d = np.zeros((nb, self.nw))
            lam_new = np.zeros((nb, self.m))
            Kkeep = [None] * nb
            delta = np.zeros(nb)
            todo = np.ones(nb, dtype=bool)
            first_try = np.ones(nb, dtype=bool)
            fail = np.zeros(nb, dtype=bool)
            while todo.any():
                t = np.nonzero(todo)[0]
                K = np.zeros((t.size, self.nw + self.m, self.nw + self.m))
                K[:, :self.nw, :self.nw] = Wl[t] + np.einsum('bi,ij->bij', Sig[t] + delta[t, None], np.eye(self.nw))
                K[:, :self.nw, self.nw:] = np.swapaxes(J[t], 1, 2)
                K[:, self.nw:, :self.nw] = J[t]
                good = _inertia_ok(K, self.nw, self.m)
                if good.any():
                    tg = t[good]
                    sol = np.linalg.solve(K[good], np.concatenate([rhs1[tg], -c[tg]], axis=1)[:, :, None])[:, :, 0]
                    d[tg] = sol[:, :self.nw]
                    lam_new[tg] = sol[:, self.nw:]
                    todo[tg] = False
                    for q_, b_ in enumerate(tg):
                        Kkeep[b_] = K[good][q_]
                tb = t[~good]
                for b in tb:                                       # W&B Alg. IC
                    if first_try[b]:
                        dl = delta_last[idx[b]]
                        delta[b] = o.delta_w_0 if dl == 0 else max(o.delta_w_min, o.kappa_w_minus * dl)
                        first_try[b] = False
                    else:
                        delta[b] *= o.kappa_w_plus_bar if delta_last[idx[b]] == 0 else o.kappa_w_plus
                    if delta[b] > o.delta_w_max:
                        fail[b] = True
                        todo[b] = False
            used = delta > 0
            delta_last[idx[used]] = delta[used]
            dzl = np.where(self.has_l, mu[idx, None] / sl - zl[idx] - zl[idx] / sl * d, 0)
            dzu = np.where(self.has_u, mu[idx, None] / su - zu[idx] + zu[idx] / su * d, 0)
            # ---- fraction to the boundary (W&B eq. 8) ----
            with np.errstate(divide='ignore', invalid='ignore'):
                a1 = np.where(self.has_l & (d < 0), -tau[idx, None] * sl / d, np.inf).min(1)
                a2 = np.where(self.has_u & (d > 0), tau[idx, None] * su / d, np.inf).min(1)
                alpha_max = np.minimum(1.0, np.minimum(a1, a2))
                b1 = np.where(self.has_l & (dzl < 0), -tau[idx, None] * zl[idx] / dzl, np.inf).min(1)
                b2 = np.where(self.has_u & (dzu < 0), -tau[idx, None] * zu[idx] / dzu, np.inf).min(1)
                alpha_z = np.minimum(1.0, np.minimum(b1, b2))
            # ---- filter line search (W&B Alg. A without SOC / restoration) ----
            phi0 = self.barrier(f_a, w[idx], mu[idx])
            th0 = np.abs(c).sum(1)
            sl_, su_ = sl, su
            gphi = g - np.where(self.has_l, mu[idx, None] / sl_, 0) + np.where(self.has_u, mu[idx, None] / su_, 0)
            dphi = np.einsum('bi,bi->b', gphi, d)
            alpha = alpha_max.copy()
            accepted = np.zeros(nb, dtype=bool)
            infeasible = np.zeros(nb, dtype=bool)
            resto = np.zeros(nb, dtype=bool)
            armijo_type = np.zeros(nb, dtype=bool)
            accepted[fail] = True                                   # nothing to search
            w_new = w[idx].copy()
            for ls in range(60):
                t = np.nonzero(~accepted)[0]
                if t.size == 0:
                    break
                wt = w[idx[t]] + alpha[t, None] * d[t]
                ft, ct = self.eval_fc(wt, self._sl(data, idx[t]))
                pht = self.barrier(ft, wt, mu[idx[t]])
                tht = np.abs(ct).sum(1)
                for q, b in enumerate(t):
                    gb = idx[b]
                    ok = np.isfinite(pht[q]) and np.isfinite(tht[q]) and tht[q] <= theta_max[gb]
                    if ok:
                        for (tf, pf) in filt[gb]:
                            if tht[q] >= tf and pht[q] - 10 * np.finfo(float).eps * abs(pf) >= pf:
                                ok = False
                                break
                    sw = False
                    if ok:
                        sw = (th0[b] <= theta_min[gb]) and (dphi[b] < 0) and \
                             (alpha[b] * (-dphi[b]) ** o.s_phi > o.delta * th0[b] ** o.s_theta)
                        rnd = 10 * np.finfo(float).eps * abs(phi0[b])       # IPOPT's round-off slack in comparisons
                        if sw:
                            ok = pht[q] - phi0[b] - rnd <= o.eta_phi * alpha[b] * dphi[b]
                        else:
                            ok = (tht[q] <= (1 - o.gamma_theta) * th0[b]) or \
                                 (pht[q] - phi0[b] - rnd <= -o.gamma_phi * th0[b])
                    if not ok and ls == 0 and o.soc and tht[q] >= th0[b] and Kkeep[b] is not None:
                        # second-order correction (W&B sec. 2.4): d_soc solves the same system with the constraint value
                        # c_soc = alpha c(x_k) + c(x_k + alpha d); up to max_soc corrections while theta drops by kappa_soc
                        c_soc = alpha[b] * c[b] + ct[q]
                        th_old = tht[q]
                        for _ in range(o.max_soc):
                            sol = np.linalg.solve(Kkeep[b], np.concatenate([rhs1[b], -c_soc]))
                            ds = sol[:self.nw]
                            slb, sub = self._slacks(w[gb:gb + 1])
                            with np.errstate(divide='ignore', invalid='ignore'):
                                a1 = np.where(self.has_l & (ds < 0), -tau[gb] * slb[0] / ds, np.inf).min()
                                a2 = np.where(self.has_u & (ds > 0), tau[gb] * sub[0] / ds, np.inf).min()
                            a_s = min(1.0, a1, a2)
                            ws = w[gb:gb + 1] + a_s * ds[None]
                            fs, cs = self.eval_fc(ws, self._sl(data, np.array([gb])))
                            phs = self.barrier(fs, ws, mu[gb:gb + 1])[0]
                            ths = np.abs(cs).sum()
                            oks = np.isfinite(phs) and np.isfinite(ths) and ths <= theta_max[gb]
                            if oks:
                                for (tf, pf) in filt[gb]:
                                    if ths >= tf and phs - 10 * np.finfo(float).eps * abs(pf) >= pf:
                                        oks = False
                                        break
                            if oks:
                                sw2 = (th0[b] <= theta_min[gb]) and (dphi[b] < 0) and \
                                      (alpha[b] * (-dphi[b]) ** o.s_phi > o.delta * th0[b] ** o.s_theta)
                                rnd = 10 * np.finfo(float).eps * abs(phi0[b])
                                if sw2:
                                    oks = phs - phi0[b] - rnd <= o.eta_phi * alpha[b] * dphi[b]
                                else:
                                    oks = (ths <= (1 - o.gamma_theta) * th0[b]) or (phs - phi0[b] - rnd <= -o.gamma_phi * th0[b])
                                if oks:
                                    ok, sw = True, sw2
                                    wt[q] = ws[0]
                                    lam_new[b] = sol[self.nw:]
                                    soc_used[gb] += 1
                                    break
                            if not (ths <= o.kappa_soc * th_old):
                                break
                            th_old = ths
                            c_soc = a_s * c_soc + cs[0]
                    if ok:
                        accepted[b] = True
                        armijo_type[b] = sw
                        w_new[b] = wt[q]
                    else:
                        alpha[b] *= o.alpha_red
                        # W&B eq. 23: below alpha_min the line search gives up and restoration is called
                        amin = o.gamma_theta
                        if dphi[b] < 0:
                            amin = min(amin, o.gamma_phi * th0[b] / (-dphi[b]))
                            if th0[b] <= theta_min[gb]:
                                amin = min(amin, o.delta * th0[b] ** o.s_theta / (-dphi[b]) ** o.s_phi)
                        if alpha[b] < o.alpha_min_frac * amin:
                            resto[b] = True
                            accepted[b] = True
            # ---- feasibility restoration (simplified W&B sec. 3.3) for the instances whose line search gave up ----
            for b in np.nonzero(resto & ~fail)[0]:
                gb = idx[b]
                filt[gb].append(((1 - o.gamma_theta) * th0[b], phi0[b] - o.gamma_phi * th0[b]))
                wr = self._restore(w[gb], self._sl(data, np.array([gb])), mu[gb], tau[gb], filt[gb], theta_max[gb])
                if isinstance(wr, str):                            # 'infeasible': IPOPT's Infeasible_Problem_Detected -> 3
                    infeasible[b] = True
                    fail[b] = True
                elif wr is None:
                    fail[b] = True
                else:
                    w_new[b] = wr
                    lam[gb] = 0.0                                  # IPOPT: constr_mult_reset_threshold = 0
                    lam_new[b] = 0.0
                    dzl[b] = 0.0
                    dzu[b] = 0.0
                    if max(zl[gb].max(), zu[gb].max()) > 1e3:      # bound_mult_reset_threshold
                        zl[gb] = np.where(self.has_l, 1.0, 0.0)
                        zu[gb] = np.where(self.has_u, 1.0, 0.0)
                    armijo_type[b] = True                          # the filter was already augmented
                    n_resto[gb] += 1
            bad = fail
            status[idx[bad]] = np.where(infeasible[bad], INFEASIBLE, RESTORATION_FAILED)
            active[idx[bad]] = False
            good = ~bad
            for b in np.nonzero(good & ~armijo_type)[0]:           # augment the filter (W&B eq. 22)
                gb = idx[b]
                filt[gb].append(((1 - o.gamma_theta) * th0[b], phi0[b] - o.gamma_phi * th0[b]))
                if len(filt[gb]) > o.max_filter:
                    filt[gb].pop(0)
            if verbose:
                print(f"      delta {delta.max():.1e} alpha {alpha.min():.2e} alpha_z {alpha_z.min():.2e} th0 {th0.max():.2e} "
                      f"dphi {dphi.min():.2e} armijo {armijo_type} resto {resto} |d| {np.abs(d).max():.2e}")
            gi = idx[good]
            w[gi] = w_new[good]
            lam[gi] = lam[gi] + alpha[good, None] * (lam_new[good] - lam[gi])
            zl[gi] = zl[gi] + alpha_z[good, None] * dzl[good]
            zu[gi] = zu[gi] + alpha_z[good, None] * dzu[good]
            # W&B eq. 16: keep z within [mu/(kappa s), kappa mu / s]
            sl, su = self._slacks(w[gi])
            zl[gi] = np.where(self.has_l, np.clip(zl[gi], mu[gi, None] / (o.kappa_sigma * sl), o.kappa_sigma * mu[gi, None] / sl), 0)
            zu[gi] = np.where(self.has_u, np.clip(zu[gi], mu[gi, None] / (o.kappa_sigma * su), o.kappa_sigma * mu[gi, None] / su), 0)
            iters[gi] += 1

        f, g, c, J, Wl = self.eval_all(w, lam, data)
        E0, dual, prim, compl = self.errors(g, c, J, lam, zl, zu, w, np.zeros(B))
        return dict(w=w, lam=lam, zl=zl, zu=zu, f=f, status=status, iters=iters, kkt=E0, n_resto=n_resto, n_soc=soc_used,
                    dual_inf=dual, prim_inf=prim, compl=compl)

    def _restore(self, w, data, mu, tau, filt, theta_max, max_it=50):
        """Feasibility restoration, simplified from W&B sec. 3.3: least-norm Newton steps on c(w) = 0
        (min |d|^2 s.t. J d = -c) with the fraction-to-the-boundary rule and an Armijo search on theta = |c|_1,
        until theta <= 0.9 theta_start and the point is acceptable to the filter.  Returns the new w or None."""
        o = self.o
        w = w[None].copy()
        lam0 = np.zeros((1, self.m))
        f, g, c, J, _ = self.eval_all(w, lam0, data)
        th_start = np.abs(c).sum()
        th = th_start
        th_ref = th
        for it_r in range(max_it):
            # ten iterations without reducing the violation by 1e-4 in total: the iterates sit at a point of locally minimal
            # infeasibility (IPOPT's restoration NLP would converge there and report Infeasible_Problem_Detected)
            if it_r % 10 == 9:
                if th > (1 - 1e-4) * th_ref and th > 1e-6:
                    return 'infeasible'
                th_ref = th
            # step of the restoration subproblem  min 1/2 |d|^2 - mu_R sum ln(slacks)  s.t.  J d = -c  (IPOPT's restoration
            # NLP keeps its iterates inside the bounds with the barrier of parameter mu_R = max(mu, |c|_inf), W&B sec. 3.3):
            # the barrier's Newton terms Sigma_R = mu_R / s^2, r_R = -mu_R / s keep the steps away from the bounds, where the
            # plain least-norm step gets stuck behind the fraction-to-the-boundary rule
            sl, su = self._slacks(w)
            mu_r = max(mu, float(np.abs(c).max())) if o.resto_barrier else 0.0
            sig = np.where(self.has_l, mu_r / sl ** 2, 0.0) + np.where(self.has_u, mu_r / su ** 2, 0.0)
            rb = -np.where(self.has_l, mu_r / sl, 0.0) + np.where(self.has_u, mu_r / su, 0.0)
            K = np.zeros((self.nw + self.m, self.nw + self.m))
            K[:self.nw, :self.nw] = np.eye(self.nw) + np.diag(sig[0])
            K[:self.nw, self.nw:] = J[0].T
            K[self.nw:, :self.nw] = J[0]
            K[self.nw:, self.nw:] = -1e-12 * np.eye(self.m)
            d = np.linalg.solve(K, np.concatenate([-rb[0], -c[0]]))[:self.nw][None]
            dmax = float(np.abs(d).max())
            if dmax <= 1e-9 and th > 1e-6:     # stationary point of the restoration problem with violated constraints
                return 'infeasible'
            sl, su = self._slacks(w)
            with np.errstate(divide='ignore', invalid='ignore'):
                a1 = np.where(self.has_l & (d < 0), -tau * sl / d, np.inf).min()
                a2 = np.where(self.has_u & (d > 0), tau * su / d, np.inf).min()
            alpha = min(1.0, a1, a2)
            ok = False
            while alpha > 1e-10:
                wt = w + alpha * d
                ft, ct = self.eval_fc(wt, data)
                tht = np.abs(ct).sum()
                if np.isfinite(tht) and tht <= (1 - 1e-4 * alpha) * th:
                    ok = True
                    break
                alpha *= 0.5
            if not ok:
                # no step length reduces the violation along the (barrier-deflected) Newton direction of the constraints: a point
                # of locally minimal infeasibility inside the box
                return 'infeasible' if th > 1e-6 else None
            w, th = wt, tht
            if th <= 0.9 * th_start and th <= theta_max:
                ph = self.barrier(ft, w, np.array([mu]))[0]
                if all(not (th >= tf and ph >= pf) for tf, pf in filt):
                    return w[0]
            f, g, c, J, _ = self.eval_all(w, lam0, data)
        return None

    # ---- reference layout helpers -----------------------------------------------------------------------------
    def to_v(self, res):
        return self.pb.join(res['X'], res['U'][:, :self.pb.Nc])

    def w_from_v(self, v):
        X, U = self.pb.split(v)
        return np.concatenate([X[:, 1:].reshape(X.shape[0], -1), U.reshape(U.shape[0], -1)], axis=1)
