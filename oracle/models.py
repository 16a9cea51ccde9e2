"""Oracle model zoo: sympy statements of the models the parity tests and the benchmark use.

TEST INFRASTRUCTURE ONLY - never imported by the product package.

The reference describes a model symbolically (CasADi SX) and obtains every derivative by AD
(`hilo_mpc/modules/dynamic_model/dynamic_model.py`); here sympy plays that role, so Jacobians and
Hessians used by the oracle are *exact symbolic* derivatives, statement-for-statement independent of
the templated forward-mode AD the HIP kernels use.

Discretisation follows `hilo_mpc/util/modeling.py:1213-1281` (explicit Runge-Kutta: stage
`k_i = f(x + h*sum_j A[i,j] k_j)`, `x+ = x + h*sum_i b_i k_i`) with the tableaux of
`modeling.py:1008-1085` and the order->tableau map of `modeling.py:1239-1250`
(1 forward Euler, 2 midpoint, 3 Kutta, 4 classic).
"""
from __future__ import annotations

import numpy as np
import sympy as sp

# ids shared with include/hilo_hip.h (HILO_MODEL_*)
MODEL_LINEAR = 0
MODEL_TOY1D = 1
MODEL_BIOREACTOR3 = 2
MODEL_CHEMOSTAT4 = 3
MODEL_PENDULUM4 = 4
MODEL_ROBOT6 = 5
MODEL_CSTR3 = 6
MODEL_CHEMOSTAT4_GP = 8

# modeling.py:1008-1085 (only the tableaux reachable through `order`)
TABLEAUX = {
    1: dict(A=[[0.]], b=[1.], c=[0.]),
    2: dict(A=[[0., 0.], [.5, 0.]], b=[0., 1.], c=[0., .5]),
    3: dict(A=[[0., 0., 0.], [.5, 0., 0.], [-1., 2., 0.]], b=[1 / 6, 2 / 3, 1 / 6], c=[0., .5, 1.]),
    4: dict(A=[[0., 0., 0., 0.], [.5, 0., 0., 0.], [0., .5, 0., 0.], [0., 0., 1., 0.]],
            b=[1 / 6, 1 / 3, 1 / 3, 1 / 6], c=[0., .5, .5, 1.]),
}


def _lam(exprs, args):
    """lambdify a list (or nested list) of scalar expressions into a batched numpy function.

    The returned callable takes arrays shaped [B, n_arg] (or [n_arg]) per argument group and returns
    an array [B, *shape(exprs)].
    """
    exprs = np.array(exprs, dtype=object)
    shape = exprs.shape
    flat = [sp.sympify(e) for e in exprs.ravel()]
    flat_args = [s for grp in args for s in grp]
    f = sp.lambdify(flat_args, flat, modules='numpy', cse=True)

    def call(*groups):
        cols = []
        B = None
        for g, grp in zip(groups, args):
            g = np.asarray(g, dtype=float)
            if g.ndim == 0:
                g = g.reshape(1)
            if g.ndim == 1:
                g = g[None, :]
            if g.shape[1] != len(grp):
                raise ValueError(f"argument group has {g.shape[1]} entries, expected {len(grp)}")
            if g.shape[0] != 1:
                B = g.shape[0] if B is None else B
            cols.append(g)
        B = 1 if B is None else B
        flat_in = [np.broadcast_to(c[:, i], (B,)) for c in cols for i in range(c.shape[1])]
        out = f(*flat_in)
        res = np.empty((B, len(flat)))
        for i, o in enumerate(out):
            res[:, i] = np.broadcast_to(np.asarray(o, dtype=float), (B,))
        return res.reshape((B,) + shape)

    return call


class OracleModel:
    """Symbolic model: x+ = f(x,u,p,dt) (discrete) or dx/dt = f(x,u,p) (continuous); y = h(x,u,p)."""

    def __init__(self, name, model_id, x, u, p, ode, meas=None, discrete=False, dt=None, z=None, alg=None):
        # z / alg: algebraic states and equations 0 = g(x, z, u, p) of a semi-explicit DAE (dynamic_model.py `set_algebraic_*`)
        self.z = list(z) if z is not None else []
        self.alg = [sp.sympify(e) for e in (alg if alg is not None else [])]
        self.name = name
        self.model_id = model_id
        self.x = list(x)
        self.u = list(u)
        self.p = list(p)
        self.dt = dt if dt is not None else sp.Symbol('dt')
        self.ode = [sp.sympify(e) for e in ode]
        self.meas = [sp.sympify(e) for e in (meas if meas is not None else [])]
        self.discrete = discrete
        self._cache = {}

    nx = property(lambda s: len(s.x))
    nu = property(lambda s: len(s.u))
    np_ = property(lambda s: len(s.p))
    ny = property(lambda s: len(s.meas))

    def discretize(self, order=4):
        """`Model.discretize('erk'|'rk4', order=...)` (dynamic_model.py:3600-3668 -> modeling.py:1213-1281)."""
        if self.discrete:
            return self
        tab = TABLEAUX[order]
        A, b = tab['A'], tab['b']
        h = self.dt
        X = sp.Matrix(self.x)
        f = sp.Matrix(self.ode)
        k = []
        for i in range(order):
            ki = sp.zeros(len(self.x), 1)
            for j in range(i):
                if A[i][j] != 0:
                    ki += sp.nsimplify(A[i][j]) * k[j] if False else A[i][j] * k[j]
            xi = X + h * ki
            k.append(f.subs(dict(zip(self.x, xi)), simultaneous=True))
        xn = X
        for i in range(order):
            if b[i] != 0:
                xn = xn + h * b[i] * k[i]
        return OracleModel(self.name + f'_erk{order}', self.model_id, self.x, self.u, self.p, list(xn),
                           self.meas, discrete=True, dt=self.dt)

    # ---- numeric callables (batched) -------------------------------------------------------
    def _get(self, key, builder):
        if key not in self._cache:
            self._cache[key] = builder()
        return self._cache[key]

    def _args(self):
        return [self.x, self.u, self.p, [self.dt]]

    def f(self, x, u, p, dt):
        return self._get('f', lambda: _lam(self.ode, self._args()))(x, u, p, dt)

    def fx(self, x, u, p, dt):
        J = sp.Matrix(self.ode).jacobian(self.x)
        return self._get('fx', lambda: _lam(J.tolist(), self._args()))(x, u, p, dt)

    def fu(self, x, u, p, dt):
        J = sp.Matrix(self.ode).jacobian(self.u) if self.nu else sp.zeros(self.nx, 0)
        if self.nu == 0:
            return np.zeros((np.atleast_2d(x).shape[0], self.nx, 0))
        return self._get('fu', lambda: _lam(J.tolist(), self._args()))(x, u, p, dt)

    def h(self, x, u, p, dt):
        if not self.meas:
            return np.atleast_2d(np.asarray(x, dtype=float))
        return self._get('h', lambda: _lam(self.meas, self._args()))(x, u, p, dt)

    def hx(self, x, u, p, dt):
        if not self.meas:
            B = np.atleast_2d(x).shape[0]
            return np.broadcast_to(np.eye(self.nx), (B, self.nx, self.nx)).copy()
        J = sp.Matrix(self.meas).jacobian(self.x)
        return self._get('hx', lambda: _lam(J.tolist(), self._args()))(x, u, p, dt)


# ---------------------------------------------------------------------------------------------
# zoo
# ---------------------------------------------------------------------------------------------
def linear2_kat():
    """2-state linear chain of tests/test_KFs.py:247-255 (continuous; the test discretises with ERK-1, dt=1)."""
    x1, x2, u, k1, k2 = sp.symbols('x_1 x_2 u k_1 k_2')
    return OracleModel('linear2', MODEL_LINEAR, [x1, x2], [u], [k1, k2],
                       [-k1 * x1 + u, k1 * x1 - k2 * x2], [x2])


def toy1d():
    """Scalar benchmark of tests/test_KFs.py:548-556: x+ = x/2 + 25 dt x/(1+x^2), y = x^2/20 (discrete)."""
    x, dt = sp.symbols('x dt')
    return OracleModel('toy1d', MODEL_TOY1D, [x], [], [], [x / 2 + 25 * dt * x / (1 + x ** 2)], [x ** 2 / 20],
                       discrete=True, dt=dt)


def bioreactor3():
    """3-state bioreactor of tests/test_KFs.py:691-712; parameters in order of first appearance
    [alpha, T_amb, mu_0, mu_1, K, Y], input D, measurements (T, cB)."""
    T, cB, cS, D = sp.symbols('T cB cS D')
    alpha, T_amb, mu_0, mu_1, K, Y = sp.symbols('alpha T_amb mu_0 mu_1 K Y')
    r = (mu_0 + mu_1 * T) * cS * cB / (K + cS)
    return OracleModel('bioreactor3', MODEL_BIOREACTOR3, [T, cB, cS], [D], [alpha, T_amb, mu_0, mu_1, K, Y],
                       [alpha * (T_amb - T), r - D * cB, -r / Y - D * cS], [T, cB])


def chemostat4():
    """CSTR-sized benchmark model: `ecoli_D1210_conti('simple')`
    (hilo_mpc/library/models.py:163-198) with the rates mu, Rs, Rfp closed by the formulas of the
    'complex' variant (models.py:143-148), the inducer factors ISF, IRF held as parameters.
    States X,S,P,I; inputs DS,DI; parameters Sf,If,ISF,IRF; measurements (X,P)."""
    X, S, P, I, DS, DI = sp.symbols('X S P I DS DI')
    Sf, If, ISF, IRF = sp.symbols('Sf If ISF IRF')
    phi = 0.407 * S / (0.108 + S + S ** 2 / 14814.0)
    mu = phi * (ISF + 0.22 * IRF / (0.22 + I))
    Rs = 2 * mu
    Rfp = phi * (0.0005 + I) / (0.022 + I)
    D = DS + DI
    return OracleModel('chemostat4', MODEL_CHEMOSTAT4, [X, S, P, I], [DS, DI], [Sf, If, ISF, IRF],
                       [mu * X - D * X, -Rs * X - D * S + DS * Sf, Rfp * X - D * P, -D * I + DI * If], [X, P])


def pendulum4():
    """Cart-pendulum of tests/test_NMPC.py:12-43 (states x,v,theta,omega; input F; all states measured)."""
    x, v, th, om, F = sp.symbols('x v theta omega F')
    M, m, l, g = 5., 1., 1., 9.81
    dv = 1. / (M + m - m * sp.cos(th)) * (m * g * sp.sin(th) - m * l * sp.sin(th) * om ** 2 + F)
    dom = 1. / l * (dv * sp.cos(th) + g * sp.sin(th))
    return OracleModel('pendulum4', MODEL_PENDULUM4, [x, v, th, om], [F], [], [v, dv, om, dom], [x, v, th, om])


def chemostat4_gp(X_train, alpha, length_scales, signal_variance=1., bias=0.):
    """`chemostat4` with the growth rate of the biomass balance replaced by a GP posterior mean over (S, I):
    `model.substitute_from(gp)` (dynamic_model.py:3040-3125; usage `nmpc_hybrid_bio.ipynb`).  The mean is written out
    term by term - bias + sum_i alpha_i sf2 exp(-1/2 sum_d (x_d - X_di)^2 / l_d^2) (inference.py:211-213,
    kernel.py:696) - like the unrolled SX expression the reference builds; Rs and Rfp keep their closed forms."""
    X, S, P, I, DS, DI = sp.symbols('X S P I DS DI')
    Sf, If, ISF, IRF = sp.symbols('Sf If ISF IRF')
    Xt = np.atleast_2d(np.asarray(X_train, dtype=float))
    al = np.asarray(alpha, dtype=float).ravel()
    ls = np.broadcast_to(np.asarray(length_scales, dtype=float), (2,))
    M = np.exp(-2 * np.log(ls))                                              # kernel.py:538-555
    sf2 = float(np.exp(2 * (np.log(signal_variance) / 2)))                   # kernel.py:127-130
    mu = sp.Float(bias) + sum(float(al[i]) * sf2 * sp.exp(-0.5 * (float(M[0]) * (S - float(Xt[0, i])) ** 2 +
                                                             float(M[1]) * (I - float(Xt[1, i])) ** 2))
                              for i in range(Xt.shape[1]))
    phi = 0.407 * S / (0.108 + S + S ** 2 / 14814.0)
    Rs = 2 * (phi * (ISF + 0.22 * IRF / (0.22 + I)))
    Rfp = phi * (0.0005 + I) / (0.022 + I)
    D = DS + DI
    return OracleModel('chemostat4_gp', MODEL_CHEMOSTAT4_GP, [X, S, P, I], [DS, DI], [Sf, If, ISF, IRF],
                       [mu * X - D * X, -Rs * X - D * S + DS * Sf, Rfp * X - D * P, -D * I + DI * If], [X, P])


def chemostat4_mu():
    """`chemostat4` with the growth rate of the biomass balance as the LAST PARAMETER `mu` - the model `substitute_from` starts
    from (dynamic_model.py:3040-3125: the label is a parameter of the model until a learned term replaces it)."""
    X, S, P, I, DS, DI, mu = sp.symbols('X S P I DS DI mu')
    Sf, If, ISF, IRF = sp.symbols('Sf If ISF IRF')
    phi = 0.407 * S / (0.108 + S + S ** 2 / 14814.0)
    Rs = 2 * (phi * (ISF + 0.22 * IRF / (0.22 + I)))
    Rfp = phi * (0.0005 + I) / (0.022 + I)
    D = DS + DI
    return OracleModel('chemostat4_mu', -1, [X, S, P, I], [DS, DI], [Sf, If, ISF, IRF, mu],
                       [mu * X - D * X, -Rs * X - D * S + DS * Sf, Rfp * X - D * P, -D * I + DI * If], [X, P])


class NumericHybridModel:
    """The filters' view of a model (nx, ny, discrete, f, fx, h, hx - oracle/kf.py) for `base` (an OracleModel whose LAST
    parameter is a learned quantity) with that parameter replaced by a function of the states that is given NUMERICALLY:
    `term(x) -> (value [B], gradient [B, nx])`, e.g. a GP posterior mean (`se_mean_term`).  The symbolic route
    (`chemostat4_gp`: 200 exponentials written out, then a Runge-Kutta map and its Jacobian by sympy) takes minutes per
    lambdify; here the chain rule is applied to numbers:
        f(x) = fb(x, mu(x)),   f_x = fb_x + fb_mu (x) dmu/dx,
    and `discretize(order)` is the explicit Runge-Kutta map of modeling.py:1213-1281 with its Jacobian propagated stage by
    stage (dk_i/dx = f_x(x_i) (I + h sum_j a_ij dk_j/dx))."""

    def __init__(self, base, term, order=None, dt=None):
        self.base, self.term, self.order = base, term, order
        self.discrete = order is not None
        self.nx, self.ny, self.nu = base.nx, base.ny, base.nu
        self.name = base.name + '_hybrid'

    def discretize(self, order=4):
        return NumericHybridModel(self.base, self.term, order=order)

    def _rhs(self, x, u, p, dt, jac):
        x = np.atleast_2d(np.asarray(x, dtype=float))
        B = x.shape[0]
        mu, dmu = self.term(x)
        pf = np.concatenate([np.broadcast_to(np.atleast_2d(p), (B, self.base.np_ - 1)), mu[:, None]], axis=1)
        f = self.base.f(x, u, pf, dt)
        if not jac:
            return f, None
        fx = self.base.fx(x, u, pf, dt)
        Jp = sp.Matrix(self.base.ode).jacobian([self.base.p[-1]])
        fmu = self.base._get('fmu', lambda: _lam(Jp.tolist(), self.base._args()))(x, u, pf, dt)      # [B, nx, 1]
        return f, fx + fmu * dmu[:, None, :]

    def _map(self, x, u, p, dt, jac):
        if not self.discrete:
            return self._rhs(x, u, p, dt, jac)
        x = np.atleast_2d(np.asarray(x, dtype=float))
        B, nx = x.shape
        tab = TABLEAUX[self.order]
        A, b = tab['A'], tab['b']
        h = float(dt)
        k, dk = [], []
        for i in range(self.order):
            xi, dxi = x.copy(), np.broadcast_to(np.eye(nx), (B, nx, nx)).copy()
            for j in range(i):
                if A[i][j] != 0:
                    xi = xi + h * A[i][j] * k[j]
                    if jac:
                        dxi = dxi + h * A[i][j] * dk[j]
            fi, Ji = self._rhs(xi, u, p, dt, jac)
            k.append(fi)
            if jac:
                dk.append(Ji @ dxi)
        xn, dxn = x.copy(), np.broadcast_to(np.eye(nx), (B, nx, nx)).copy()
        for i in range(self.order):
            xn = xn + h * b[i] * k[i]
            if jac:
                dxn = dxn + h * b[i] * dk[i]
        return xn, dxn

    def f(self, x, u, p, dt):
        return self._map(x, u, p, dt, False)[0]

    def fx(self, x, u, p, dt):
        return self._map(x, u, p, dt, True)[1]

    def h(self, x, u, p, dt):
        B = np.atleast_2d(x).shape[0]
        pf = np.concatenate([np.broadcast_to(np.atleast_2d(p), (B, self.base.np_ - 1)), np.zeros((B, 1))], axis=1)
        return self.base.h(x, u, pf, dt)

    def hx(self, x, u, p, dt):
        B = np.atleast_2d(x).shape[0]
        pf = np.concatenate([np.broadcast_to(np.atleast_2d(p), (B, self.base.np_ - 1)), np.zeros((B, 1))], axis=1)
        return self.base.hx(x, u, pf, dt)


def se_mean_term(X_train, alpha, length_scales, signal_variance, state_index, bias=0.):
    """Posterior mean of a GP with the squared-exponential kernel (ARD) over the states `state_index`, and its gradient with
    respect to ALL states:  m(x) = bias + sum_i alpha_i sf2 exp(-1/2 sum_d (x_d - X_di)^2 / l_d^2)  (inference.py:211-213,
    kernel.py:696)."""
    Xt = np.atleast_2d(np.asarray(X_train, dtype=float))
    al = np.asarray(alpha, dtype=float).ravel()
    M = np.exp(-2 * np.log(np.broadcast_to(np.asarray(length_scales, dtype=float), (Xt.shape[0],))))
    sf2 = float(np.exp(2 * (np.log(signal_variance) / 2)))
    idx = list(state_index)

    def term(x):
        x = np.atleast_2d(np.asarray(x, dtype=float))
        d = x[:, idx, None] - Xt[None, :, :]                                    # [B, D, n]
        kv = sf2 * np.exp(-0.5 * np.einsum('d,bdn->bn', M, d * d)) * al[None, :]
        g = np.zeros_like(x)
        g[:, idx] = -np.einsum('bn,bdn->bd', kv, d) * M[None, :]
        return bias + kv.sum(1), g
    return term


def pendulum4_dae():
    """The DAE of the reference's own NMPC test (tests/test_NMPC.py:1866-1911): the cart-pendulum with the height of the
    pendulum tip as algebraic state, 0 = h + l cos(theta) - y."""
    m = pendulum4()
    y = sp.Symbol('y')
    return OracleModel('pendulum4_dae', -1, m.x, m.u, [], m.ode, m.meas, z=[y], alg=[0.5 + 1.0 * sp.cos(m.x[2]) - y])


def chemostat4_dae():
    """chemostat4 with the growth rate as an ALGEBRAIC state that feeds back into the biomass and substrate balances:
    0 = mu - phi(S) (ISF + 0.22 IRF / (0.22 + I)).  Eliminating mu gives chemostat4 itself - the DAE oracle and the product's
    eliminated form must both land on the ODE problem's solution."""
    X, S, P, I, DS, DI, mu = sp.symbols('X S P I DS DI mu')
    Sf, If, ISF, IRF = sp.symbols('Sf If ISF IRF')
    phi = 0.407 * S / (0.108 + S + S ** 2 / 14814.0)
    Rfp = phi * (0.0005 + I) / (0.022 + I)
    D = DS + DI
    return OracleModel('chemostat4_dae', -1, [X, S, P, I], [DS, DI], [Sf, If, ISF, IRF],
                       [mu * X - D * X, -2 * mu * X - D * S + DS * Sf, Rfp * X - D * P, -D * I + DI * If], [X, P],
                       z=[mu], alg=[mu - phi * (ISF + 0.22 * IRF / (0.22 + I))])


def robot6_dae():
    """C5's DAE variant: the robot with the squared speed as algebraic state, 0 = z - (vx^2 + vy^2), which the
    dynamics do not use (a constraint / output quantity, the pendulum pattern)."""
    m = robot6()
    z = sp.Symbol('z')
    return OracleModel('robot6_dae', -1, m.x, m.u, [], m.ode, m.meas, z=[z], alg=[z - (m.x[1] ** 2 + m.x[3] ** 2)])


def robot6():
    """Planar mobile robot with heading for the path-following configuration C5 (the reference holds no
    6-state model - pattern of the point mass in tests/test_NMPC.py:742-775 and path_following_mpc.ipynb cell 3):
    states px,vx,py,vy,psi,omega; inputs a (acceleration along the heading) and alpha; measurements (px, py)."""
    px, vx, py, vy, psi, om, a, al = sp.symbols('px vx py vy psi omega a alpha')
    return OracleModel('robot6', MODEL_ROBOT6, [px, vx, py, vy, psi, om], [a, al], [],
                       [vx, a * sp.cos(psi), vy, a * sp.sin(psi), om, al], [px, py])


def cstr3():
    """Reversible exothermic reaction A <-> B in a cooled CSTR, `docs/docsource/examples/CSTR_Example.ipynb` cell 6
    (`true_plant_model`, constants of cell 4): states C_A, C_B, T; input Q; measurement r (the reaction rate)."""
    CA, CB, T, Q = sp.symbols('C_A C_B T Q')
    T_0, tau, k_A, k_B, E_A, E_B, R, dH, rho, Cp, C_A_0, V = 400, 60, 5000, 1e6, 1e4, 1.5e4, 1.987, -5000, 1, 1000, 1, 100
    r = k_A * sp.exp((-E_A) / (R * T)) * CA - k_B * sp.exp((-E_B) / (R * T)) * CB
    dCA = sp.Rational(1, 1) / tau * (C_A_0 - CA) - r
    dCB = -sp.Rational(1, 1) / tau * CB + r
    dT = -(dH * r) / (rho * Cp) + sp.Rational(1, 1) / tau * (T_0 - T) + Q / (rho * Cp * V)
    return OracleModel('cstr3', MODEL_CSTR3, [CA, CB, T], [Q], [], [dCA, dCB, dT], [r])


def racecar2():
    """The model of the reference's minimum-time test (tests/test_NMPC.py:2706-2735): position and speed of a car, p' = v,
    v' = u - v."""
    pp, v, u = sp.symbols('p v u')
    return OracleModel('racecar2', -1, [pp, v], [u], [], [v, u - v], [pp, v])


ZOO = {
    'racecar2': racecar2,
    'cstr3': cstr3,
    'linear2': linear2_kat,
    'toy1d': toy1d,
    'bioreactor3': bioreactor3,
    'chemostat4': chemostat4,
    'pendulum4': pendulum4,
    'robot6': robot6,
    'pendulum4_dae': pendulum4_dae,
    'robot6_dae': robot6_dae,
    'chemostat4_dae': chemostat4_dae,
}


def get(name):
    return ZOO[name]()
