"""Oracle: Kalman / extended / unscented Kalman filter steps, batched numpy.

TEST INFRASTRUCTURE ONLY - never imported by the product package.

Follows `hilo_mpc/modules/estimator/kf.py` (reference v1.1.0):

* predict   kf.py:71-133   x- = f(x,u,p); P- = F P F^T + Q (discrete, :95-96) or the augmented ODE
                           [x; vec P]' = [f; F P + P F^T + Q] integrated over dt (continuous, :97-110,124)
* update    kf.py:135-186  P_xy = P H^T, P_yy = H P H^T + R, K = (P_yy^T \\ P_xy^T)^T (:169-177),
                           x+ = x + K (y - h(x)), P+ = P - K P_yy K^T (:179-180)  (not Joseph form)
* step      kf.py:207-265  update(predict(.)) - predict first, then update (:258-259)
* EKF       kf.py:215-217  F evaluated at the *prior* state (:91), H at the *predicted* state (:164)
* UKF       kf.py:486-604  lambda, gamma, weights (:493-500); sigma points x, x +/- gamma*S[:,k] with
                           S = chol(P) the UPPER factor (CasADi convention) and its *columns* (:503,522-527);
                           update re-uses the propagated points (:580-592)

Data layout: every array carries a leading batch axis.  The packed tile `[x | P]` is `[B, nx, nx+1]`
(`ca.horzcat(x, P)`, kf.py:129-133); the UKF prediction tile `[x | P | X]` is `[B, nx, 1+nx+(2nx+1)]`
(kf.py:550-554).
"""
from __future__ import annotations

import numpy as np
from scipy.integrate import solve_ivp


def _b(a, B=None, nd=2):
    a = np.asarray(a, dtype=float)
    while a.ndim < nd:
        a = a[None]
    if B is not None and a.shape[0] == 1 and B != 1:
        a = np.broadcast_to(a, (B,) + a.shape[1:])
    return a


def as_cov(v, n):
    """`_Estimator` Q/R setters (estimator/base.py:105-125): scalar or vector -> diagonal, matrix -> as is."""
    v = np.asarray(v, dtype=float)
    if v.ndim == 0:
        return np.eye(n) * float(v)
    if v.ndim == 1:
        if v.size == 1 and n > 1:
            return np.eye(n) * float(v[0])
        return np.diag(v)
    return v


def pack(x, P):
    x = _b(x)
    P = _b(P, x.shape[0], 3)
    return np.concatenate([x[:, :, None], P], axis=2)


def unpack(xP):
    xP = _b(xP, nd=3)
    return xP[:, :, 0], xP[:, :, 1:]


# ---------------------------------------------------------------------------------------------
# KF / EKF
# ---------------------------------------------------------------------------------------------
def _integrate_continuous(model, x, P, u, p, Q, dt, rtol=1e-11, atol=1e-12):
    """kf.py:97-110: integrate [x; vec P] with Pdot = F(x) P + P F(x)^T + Q over one sampling interval.
    (The reference hands this to CVODES; the oracle uses scipy's DOP853 at tight tolerance.)"""
    B, nx = x.shape
    xo = np.empty_like(x)
    Po = np.empty_like(P)
    for b in range(B):
        ub, pb, Qb = u[b], p[b], Q[b if Q.shape[0] > 1 else 0]

        def rhs(t, s):
            xs = s[:nx]
            Ps = s[nx:].reshape(nx, nx)
            F = model.fx(xs, ub, pb, dt)[0]
            dP = F @ Ps + Ps @ F.T + Qb
            return np.concatenate([model.f(xs, ub, pb, dt)[0], dP.ravel()])

        sol = solve_ivp(rhs, (0., float(dt)), np.concatenate([x[b], P[b].ravel()]), method='DOP853',
                        rtol=rtol, atol=atol)
        xo[b] = sol.y[:nx, -1]
        Po[b] = sol.y[nx:, -1].reshape(nx, nx)
    return xo, Po


def kf_predict(model, xP, u, p, Q, dt):
    """`_KalmanFilter._setup_predict` (kf.py:71-133). Works for KF (linear model) and EKF alike:
    F is the model Jacobian at the prior state, which for a linear model is the state matrix."""
    x, P = unpack(xP)
    B = x.shape[0]
    u = _b(u, B)
    p = _b(p, B)
    Q = _b(as_cov(Q, model.nx) if np.ndim(Q) < 3 else Q, nd=3)
    if model.discrete:
        F = model.fx(x, u, p, dt)
        xn = model.f(x, u, p, dt)
        Pn = F @ P @ np.swapaxes(F, 1, 2) + Q
    else:
        xn, Pn = _integrate_continuous(model, x, P, u, p, Q, dt)
    return pack(xn, Pn)


def gain_update(x, P, P_xy, P_yy, y, y_pred):
    """kf.py:177-180 / :595-598 (shared by KF, EKF and UKF)."""
    K = np.swapaxes(np.linalg.solve(np.swapaxes(P_yy, 1, 2), np.swapaxes(P_xy, 1, 2)), 1, 2)
    x_up = x + (K @ (y - y_pred)[:, :, None])[:, :, 0]
    P_up = P - K @ P_yy @ np.swapaxes(K, 1, 2)
    return x_up, P_up


def kf_update(model, xP, y, u, p, R, dt):
    """`_KalmanFilter._setup_update` (kf.py:135-186). Returns ([x+|P+], y_pred)."""
    x, P = unpack(xP)
    B = x.shape[0]
    u = _b(u, B)
    p = _b(p, B)
    ny = model.ny if model.ny else model.nx
    y = _b(y, B)
    R = _b(as_cov(R, ny) if np.ndim(R) < 3 else R, nd=3)
    H = model.hx(x, u, p, dt)
    y_pred = model.h(x, u, p, dt)
    P_xy = P @ np.swapaxes(H, 1, 2)
    P_yy = H @ P @ np.swapaxes(H, 1, 2) + R
    x_up, P_up = gain_update(x, P, P_xy, P_yy, y, y_pred)
    return pack(x_up, P_up), y_pred


def kf_step(model, xP, y, u, p, Q, R, dt):
    """One `estimate()` call: update(predict(.)) (kf.py:258-265)."""
    return kf_update(model, kf_predict(model, xP, u, p, Q, dt), y, u, p, R, dt)


# ---------------------------------------------------------------------------------------------
# UKF
# ---------------------------------------------------------------------------------------------
def ukf_weights(nx, alpha=1e-3, beta=2., kappa=0.):
    """kf.py:486-500. Returns (gamma, W[2, 2nx+1]) with W[0]=mean weights, W[1]=covariance weights."""
    lam = alpha ** 2 * (nx + kappa) - nx
    gamma = np.sqrt(nx + lam)
    W = np.zeros((2, 2 * nx + 1))
    W[0, 0] = lam / (nx + lam)
    W[1, 0] = lam / (nx + lam) + 1 - alpha ** 2 + beta
    W[:, 1:] = 1 / (2 * (nx + lam))
    return gamma, W


def chol_upper(P):
    """`ca.chol` returns the upper factor R with R^T R = P (kf.py:503)."""
    return np.swapaxes(np.linalg.cholesky(P), -1, -2)


def _propagate(model, X, u, p, dt, rtol=1e-11, atol=1e-12):
    """Model step for every sigma point (kf.py:539-540). X is [B, nx, ns]."""
    B, nx, ns = X.shape
    out = np.empty_like(X)
    for k in range(ns):
        if model.discrete:
            out[:, :, k] = model.f(X[:, :, k], u, p, dt)
        else:
            for b in range(B):
                sol = solve_ivp(lambda t, s: model.f(s, u[b], p[b], dt)[0], (0., float(dt)), X[b, :, k],
                                method='DOP853', rtol=rtol, atol=atol)
                out[b, :, k] = sol.y[:, -1]
    return out


def ukf_predict(model, xP, u, p, Q, dt, alpha=1e-3, beta=2., kappa=0.):
    """`UnscentedKalmanFilter._setup_predict` (kf.py:505-554). Returns [x-|P-|X] as [B, nx, 1+nx+2nx+1]."""
    x, P = unpack(xP)
    B, nx = x.shape
    u = _b(u, B)
    p = _b(p, B)
    Q = _b(as_cov(Q, nx) if np.ndim(Q) < 3 else Q, nd=3)
    gamma, W = ukf_weights(nx, alpha, beta, kappa)
    S = chol_upper(P)
    X = np.empty((B, nx, 2 * nx + 1))
    X[:, :, 0] = x
    for k in range(nx):
        X[:, :, 1 + k] = x + gamma * S[:, :, k]
        X[:, :, 1 + nx + k] = x - gamma * S[:, :, k]
    X = _propagate(model, X, u, p, dt)
    # sequential accumulation in the reference's order (kf.py:542-548): with the default alpha = 1e-3 the
    # centre weight is ~ -1e6 and the sums cancel by six digits, so the order is part of the result
    x_pred = np.zeros((B, nx))
    for k in range(2 * nx + 1):
        x_pred = x_pred + W[0, k] * X[:, :, k]
    P_pred = np.broadcast_to(Q, (B, nx, nx)).copy()
    for k in range(2 * nx + 1):
        d = X[:, :, k] - x_pred
        P_pred = P_pred + (W[1, k] * d)[:, :, None] @ d[:, None, :]
    return np.concatenate([x_pred[:, :, None], P_pred, X], axis=2)


def ukf_update(model, xPX, y, u, p, R, dt, alpha=1e-3, beta=2., kappa=0.):
    """`UnscentedKalmanFilter._setup_update` (kf.py:556-604). Returns ([x+|P+], y_pred)."""
    xPX = _b(xPX, nd=3)
    nx = xPX.shape[1]
    x, P, X = xPX[:, :, 0], xPX[:, :, 1:1 + nx], xPX[:, :, 1 + nx:]
    B = x.shape[0]
    u = _b(u, B)
    p = _b(p, B)
    ny = model.ny if model.ny else nx
    y = _b(y, B)
    R = _b(as_cov(R, ny) if np.ndim(R) < 3 else R, nd=3)
    _, W = ukf_weights(nx, alpha, beta, kappa)
    Y = np.stack([model.h(X[:, :, k], u, p, dt) for k in range(2 * nx + 1)], axis=2)
    y_pred = np.zeros((B, ny))
    for k in range(2 * nx + 1):                                   # kf.py:583-585
        y_pred = y_pred + W[0, k] * Y[:, :, k]
    P_xy = np.zeros((B, nx, ny))
    P_yy = np.broadcast_to(R, (B, ny, ny)).copy()
    for k in range(2 * nx + 1):                                   # kf.py:588-592
        dx = X[:, :, k] - x
        dy = Y[:, :, k] - y_pred
        P_xy = P_xy + (W[1, k] * dx)[:, :, None] @ dy[:, None, :]
        P_yy = P_yy + (W[1, k] * dy)[:, :, None] @ dy[:, None, :]
    x_up, P_up = gain_update(x, P, P_xy, P_yy, y, y_pred)
    return pack(x_up, P_up), y_pred


def ukf_step(model, xP, y, u, p, Q, R, dt, alpha=1e-3, beta=2., kappa=0.):
    return ukf_update(model, ukf_predict(model, xP, u, p, Q, dt, alpha, beta, kappa), y, u, p, R, dt,
                      alpha, beta, kappa)
