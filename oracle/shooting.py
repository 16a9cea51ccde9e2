"""Oracle: shooting map x+ = Phi(x, u, p) of a (pre-)discretised model with exact first and second derivatives.

TEST INFRASTRUCTURE ONLY - never imported by the product package.

The reference obtains Phi symbolically (`Model.discretize('rk4')` -> `RungeKutta._explicit`,
hilo_mpc/util/modeling.py:1213-1281) and lets CasADi differentiate it twice inside `ca.nlpsol` (exact Hessian,
SURVEY 2.2 K2).  sympy cannot differentiate the nested RK4 expression in reasonable time, so the oracle
differentiates the *continuous* right-hand side symbolically (f, f_w, f_ww with w = (x, u)) and pushes first and
second derivatives through the Runge-Kutta stages with the chain rule - tensor algebra in numpy, formulated
independently of the univariate-Taylor/polarisation scheme the HIP kernels use.
"""
from __future__ import annotations

import numpy as np
import sympy as sp

from .models import TABLEAUX, _lam


class ShootingMap:
    def __init__(self, model, order=4, n_sub=1):
        self.m = model
        self.order = order
        self.n_sub = n_sub
        self.nx, self.nu = model.nx, model.nu
        self.nz = self.nx + self.nu
        w = model.x + model.u
        args = [model.x, model.u, model.p, [model.dt]]
        F = sp.Matrix(model.ode)
        self._f = _lam(list(F), args)
        self._fw = _lam(F.jacobian(w).tolist(), args)
        H = [[[sp.diff(F[m], a, b) for b in w] for a in w] for m in range(self.nx)]
        self._fww = _lam(H, args)

    def _rhs(self, x, u, p, dt):
        return self._f(x, u, p, dt), self._fw(x, u, p, dt), self._fww(x, u, p, dt)

    def _erk(self, x, dx, ddx, u, p, h):
        """One ERK step for value x [B,nx], first derivative dx [B,nx,nz], second ddx [B,nx,nz,nz] w.r.t. the
        interval's z = (x_k, u_k)."""
        tab = TABLEAUX[self.order]
        A, b = tab['A'], tab['b']
        B, nx, nu, nz = x.shape[0], self.nx, self.nu, self.nz
        du = np.zeros((B, nu, nz))
        du[:, :, nx:] = np.eye(nu)
        k, dk, ddk = [], [], []
        for i in range(self.order):
            xi, dxi, ddxi = x.copy(), dx.copy(), ddx.copy()
            for j in range(i):
                if A[i][j] != 0:
                    xi += h * A[i][j] * k[j]
                    dxi += h * A[i][j] * dk[j]
                    ddxi += h * A[i][j] * ddk[j]
            f, fw, fww = self._rhs(xi, u, p, h)
            dW = np.concatenate([dxi, du], axis=1)                       # [B, nz(w), nz(z)]
            ddW = np.concatenate([ddxi, np.zeros((B, nu, nz, nz))], axis=1)
            k.append(f)
            dk.append(np.einsum('bma,baz->bmz', fw, dW))
            ddk.append(np.einsum('bmac,baz,bcy->bmzy', fww, dW, dW) + np.einsum('bma,bazy->bmzy', fw, ddW))
        xn, dxn, ddxn = x.copy(), dx.copy(), ddx.copy()
        for i in range(self.order):
            if b[i] != 0:
                xn += h * b[i] * k[i]
                dxn += h * b[i] * dk[i]
                ddxn += h * b[i] * ddk[i]
        return xn, dxn, ddxn

    def __call__(self, x, u, p, dt):
        """Returns Phi [B,nx], dPhi/dz [B,nx,nz], d2Phi/dz2 [B,nx,nz,nz] with z = (x,u)."""
        x = np.atleast_2d(np.asarray(x, dtype=float))
        B = x.shape[0]
        u = np.broadcast_to(np.atleast_2d(np.asarray(u, dtype=float)), (B, self.nu))
        p = np.broadcast_to(np.atleast_2d(np.asarray(p, dtype=float)), (B, self.m.np_))
        nx, nz = self.nx, self.nz
        if self.m.discrete:
            f, fw, fww = self._rhs(x, u, p, dt)
            return f, fw, fww
        dx = np.zeros((B, nx, nz))
        dx[:, :, :nx] = np.eye(nx)
        ddx = np.zeros((B, nx, nz, nz))
        h = dt / self.n_sub
        xc = x.copy()
        for _ in range(self.n_sub):
            xc, dx, ddx = self._erk(xc, dx, ddx, u, p, h)
        return xc, dx, ddx

    def value(self, x, u, p, dt):
        x = np.atleast_2d(np.asarray(x, dtype=float))
        B = x.shape[0]
        u = np.broadcast_to(np.atleast_2d(np.asarray(u, dtype=float)), (B, self.nu))
        p = np.broadcast_to(np.atleast_2d(np.asarray(p, dtype=float)), (B, self.m.np_))
        if self.m.discrete:
            return self._f(x, u, p, dt)
        tab = TABLEAUX[self.order]
        A, b = tab['A'], tab['b']
        h = dt / self.n_sub
        xc = x.copy()
        for _ in range(self.n_sub):
            k = []
            for i in range(self.order):
                xi = xc.copy()
                for j in range(i):
                    if A[i][j] != 0:
                        xi = xi + h * A[i][j] * k[j]
                k.append(self._f(xi, u, p, h))
            for i in range(self.order):
                if b[i] != 0:
                    xc = xc + h * b[i] * k[i]
        return xc
