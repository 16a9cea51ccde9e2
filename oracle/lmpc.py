"""Oracle: linear MPC - the reference's QP assembly + a dense QP solver, numpy.

TEST INFRASTRUCTURE ONLY - never imported by the product package.   PARITY UNPINNED: tests/test_LMPC.py asserts no
numbers and qpOASES lives in the un-installable `casadi` dependency.  The QP solver below (Mehrotra predictor-corrector
followed by an exact active-set polish) is cross-checked against scipy in tests/test_oracle_lmpc.py.

Restates `LMPC.setup` / `LMPC.optimize` (hilo_mpc/modules/controller/mpc.py:2143-2305, 2307-2394):
  v   = [x_0..x_N | u_0..u_{N-1}]                                  (mpc.py:2221-2231)
  H   = blkdiag(I_N (x) Q, P, I_N (x) R)                           (mpc.py:2252-2256)      cost = 1/2 v^T H v (conic: Q6)
  Aeq = [kron(I_N, A) | 0] + kron(shift, -I_nx)  |  B-block        (mpc.py:2209-2245),  Aeq v = 0
        B-block = kron(B, I_N)  without time-varying parameters    (mpc.py:2243)  <- Q5: row order disagrees with the
                  diagcat(B, ..., B) with them                     (mpc.py:2236-2240)    state block when nx > 1
  bounds tiled (mpc.py:2259-2266), scaled by the scaling vectors (mpc.py:2071-2075) - the model matrices are NOT scaled;
  x_0 pinned through its bounds (mpc.py:2361-2362); returns v[u_ind[0]] * u_scaling (mpc.py:2377).
"""
from __future__ import annotations

import numpy as np

INF = np.inf


class LmpcProblem:
    def __init__(self, A, B, N, Q=None, R=None, P=None, x_lb=None, x_ub=None, u_lb=None, u_ub=None,
                 x_scaling=None, u_scaling=None, kron_bug=True):
        A = np.atleast_2d(np.asarray(A, dtype=float))
        B = np.atleast_2d(np.asarray(B, dtype=float))
        if B.shape[0] != A.shape[0]:
            B = B.T
        nx, nu = A.shape[0], B.shape[1]
        self.A, self.B, self.N, self.nx, self.nu = A, B, int(N), nx, nu
        Q = np.zeros((nx, nx)) if Q is None else np.atleast_2d(np.asarray(Q, dtype=float))       # mpc.py:2188-2193
        P = np.zeros((nx, nx)) if P is None else np.atleast_2d(np.asarray(P, dtype=float))
        R = np.zeros((nu, nu)) if R is None else np.atleast_2d(np.asarray(R, dtype=float))
        self.sx = np.ones(nx) if x_scaling is None else np.asarray(x_scaling, dtype=float)
        self.su = np.ones(nu) if u_scaling is None else np.asarray(u_scaling, dtype=float)
        N = self.N
        Abar1 = np.hstack([np.kron(np.eye(N), A), np.zeros((N * nx, nx))])                      # mpc.py:2209-2213
        aux2 = np.zeros((N, N + 1))
        for i in range(N):
            aux2[i, i + 1] = -1
        Abar2 = np.kron(aux2, np.eye(nx))                                                         # mpc.py:2233
        Abar3 = np.kron(B, np.eye(N)) if kron_bug else np.kron(np.eye(N), B)                     # mpc.py:2243 / :2236-2240
        self.Aeq = np.hstack([Abar1 + Abar2, Abar3])
        self.beq = np.zeros(N * nx)
        self.H = np.zeros(((N + 1) * nx + N * nu,) * 2)
        self.H[:N * nx, :N * nx] = np.kron(np.eye(N), Q)
        self.H[N * nx:(N + 1) * nx, N * nx:(N + 1) * nx] = P
        self.H[(N + 1) * nx:, (N + 1) * nx:] = np.kron(np.eye(N), R)
        self.g = np.zeros(self.H.shape[0])
        xl = (np.full(nx, -INF) if x_lb is None else np.asarray(x_lb, dtype=float)) / self.sx
        xu = (np.full(nx, INF) if x_ub is None else np.asarray(x_ub, dtype=float)) / self.sx
        ul = (np.full(nu, -INF) if u_lb is None else np.asarray(u_lb, dtype=float)) / self.su
        uu = (np.full(nu, INF) if u_ub is None else np.asarray(u_ub, dtype=float)) / self.su
        self.v_lb = np.concatenate([np.tile(xl, N + 1), np.tile(ul, N)])
        self.v_ub = np.concatenate([np.tile(xu, N + 1), np.tile(uu, N)])
        self.x_ind = [list(range(k * nx, (k + 1) * nx)) for k in range(N + 1)]                  # mpc.py:2221-2231
        self.u_ind = [list(range((N + 1) * nx + k * nu, (N + 1) * nx + (k + 1) * nu)) for k in range(N)]
        self.n_v = self.H.shape[0]

    def bounds_for(self, x0):
        lb, ub = self.v_lb.copy(), self.v_ub.copy()
        lb[self.x_ind[0]] = ub[self.x_ind[0]] = np.asarray(x0, dtype=float) / self.sx            # mpc.py:2361-2362
        return lb, ub


def solve_qp(H, g, A, b, lb, ub, tol=1e-10, max_iter=100, reg=1e-11):
    """min 1/2 x^T H x + g^T x  s.t.  A x = b,  lb <= x <= ub  (H psd).  Mehrotra predictor-corrector on the free
    variables (fixed ones, lb == ub, are substituted), then an active-set polish: the bound multipliers of the
    interior-point solution identify the active set and the resulting equality-constrained KKT system is solved
    exactly.  Returns dict(x, y, z, f, status, iters) with z = zu - zl (CasADi's lam_x sign)."""
    H, A = np.asarray(H, dtype=float), np.asarray(A, dtype=float)
    n, m = H.shape[0], A.shape[0]
    fixed = lb == ub
    free = ~fixed
    xf = np.where(fixed, lb, 0.0)
    Hf, Af = H[np.ix_(free, free)], A[:, free]
    gf = g[free] + H[np.ix_(free, fixed)] @ xf[fixed]
    bf = b - A[:, fixed] @ xf[fixed]
    l, u = lb[free], ub[free]
    hl, hu = np.isfinite(l), np.isfinite(u)
    nf = int(free.sum())
    x = np.zeros(nf)
    both = hl & hu
    x = np.where(both, .5 * (l + u), x)
    x = np.where(hl & ~hu, np.maximum(x, l + 1.), x)
    x = np.where(hu & ~hl, np.minimum(x, u - 1.), x)
    y = np.zeros(m)
    zl, zu = np.where(hl, 1., 0.), np.where(hu, 1., 0.)
    nb = max(1, int(hl.sum() + hu.sum()))
    status, it, phi_min = 5, 0, INF
    for it in range(max_iter):
      with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
          sl, su = np.where(hl, x - l, 1.), np.where(hu, u - x, 1.)
          rd = Hf @ x + gf + Af.T @ y - zl + zu
          rp = Af @ x - bf
          mu = (np.sum(sl * zl * hl) + np.sum(su * zu * hu)) / nb
          phi = max(np.abs(rd).max(initial=0) / (1 + np.abs(gf).max(initial=0)), np.abs(rp).max(initial=0), mu)
          if phi <= tol:
              status = 1
              break
          # infeasible QP: the termination rule of OOQP (Gertz & Wright, ACM TOMS 29, 2003) - the merit has grown to 1e4 times
          # its smallest value so far (the reference's qpOASES reports infeasibility; this iteration would diverge instead)
          phi_min = min(phi_min, phi)
          if phi >= 1e4 * phi_min:
              status = 3
              break
          M = Hf + np.diag(np.where(hl, zl / sl, 0) + np.where(hu, zu / su, 0) + reg)
          L = np.linalg.cholesky(M)
          X = np.linalg.solve(L, Af.T)
          S = X.T @ X + reg * np.eye(m)
          Ls = np.linalg.cholesky(S)

          def newton(r1):
              t = np.linalg.solve(L, r1)
              dy = np.linalg.solve(Ls.T, np.linalg.solve(Ls, X.T @ t + rp))
              dx = np.linalg.solve(L.T, t - X @ dy)
              return dx, dy

          base = -(Hf @ x + gf + Af.T @ y)
          dx, dy = newton(base)                                                  # predictor (sigma = 0)
          dzl = np.where(hl, -zl - zl / sl * dx, 0)
          dzu = np.where(hu, -zu + zu / su * dx, 0)

          def steps(dx, dzl, dzu, tau):
              with np.errstate(divide='ignore', invalid='ignore'):
                  ap = min(1., np.where(hl & (dx < 0), -tau * sl / dx, INF).min(initial=INF),
                           np.where(hu & (dx > 0), tau * su / dx, INF).min(initial=INF))
                  ad = min(1., np.where(hl & (dzl < 0), -tau * zl / dzl, INF).min(initial=INF),
                           np.where(hu & (dzu < 0), -tau * zu / dzu, INF).min(initial=INF))
              return ap, ad

          ap, ad = steps(dx, dzl, dzu, 1.)
          mu_aff = (np.sum((sl + ap * dx) * (zl + ad * dzl) * hl) + np.sum((su - ap * dx) * (zu + ad * dzu) * hu)) / nb
          sigma = (mu_aff / mu) ** 3 if mu > 0 else 0.
          cl, cu = dx * dzl, -dx * dzu                                           # second-order terms
          r1 = base + np.where(hl, (sigma * mu - cl) / sl, 0) - np.where(hu, (sigma * mu - cu) / su, 0)
          dx, dy = newton(r1)
          dzl = np.where(hl, (sigma * mu - cl) / sl - zl - zl / sl * dx, 0)
          dzu = np.where(hu, (sigma * mu - cu) / su - zu + zu / su * dx, 0)
          ap, ad = steps(dx, dzl, dzu, max(.995, 1 - mu))
          x = x + ap * dx
          y = y + ad * dy
          zl, zu = zl + ad * dzl, zu + ad * dzu
    # ---- active-set polish ----
    act_l = hl & (zl > 1e-6) & (x - l < 1e-6)
    act_u = hu & (zu > 1e-6) & (u - x < 1e-6)
    idx = np.nonzero(act_l | act_u)[0]
    E = np.zeros((idx.size, nf))
    E[np.arange(idx.size), idx] = 1.
    e = np.where(act_l, l, u)[idx]
    K = np.block([[Hf + reg * 0 * np.eye(nf), Af.T, E.T], [Af, np.zeros((m, m)), np.zeros((m, idx.size))],
                  [E, np.zeros((idx.size, m)), np.zeros((idx.size, idx.size))]])
    try:
        sol = np.linalg.solve(K, np.concatenate([-gf, bf, e]))
        xp, yp, wp = sol[:nf], sol[nf:nf + m], sol[nf + m:]
        zp = np.zeros(nf)
        zp[idx] = wp
        ok = np.all(xp >= l - 1e-9) and np.all(xp <= u + 1e-9) and np.all(zp[act_l] <= 1e-9) and np.all(zp[act_u] >= -1e-9)
        if ok and np.abs(xp - x).max() < 1e-4:
            x, y = xp, yp
            zl, zu = np.where(act_l, -zp, 0.), np.where(act_u, zp, 0.)
    except np.linalg.LinAlgError:
        pass
    xfull = xf.copy()
    xfull[free] = x
    z = np.zeros(n)
    z[free] = zu - zl
    z[fixed] = -(H @ xfull + g + A.T @ y)[fixed]
    return dict(x=xfull, y=y, z=z, f=.5 * xfull @ H @ xfull + g @ xfull, status=status, iters=it)


def lmpc_optimize(pb: LmpcProblem, x0):
    """`LMPC.optimize` for a batch of measured states x0 [B, nx] -> (u [B, nu], v [B, n_v], f [B], status [B])."""
    x0 = np.atleast_2d(np.asarray(x0, dtype=float))
    out = []
    for b in range(x0.shape[0]):
        lb, ub = pb.bounds_for(x0[b])
        out.append(solve_qp(pb.H, pb.g, pb.Aeq, pb.beq, lb, ub))
    v = np.stack([o['x'] for o in out])
    return dict(u=v[:, pb.u_ind[0]] * pb.su, v=v, f=np.array([o['f'] for o in out]),
                status=np.array([o['status'] for o in out], dtype=np.int32),
                iters=np.array([o['iters'] for o in out], dtype=np.int32),
                lam_a=np.stack([o['y'] for o in out]), lam_x=np.stack([o['z'] for o in out]))
