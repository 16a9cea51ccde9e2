"""The continuous models of the device zoo written as expressions - statement by statement what csrc/hilo_models.h holds as
hand-written functors.  Two uses: tools/gen_model_sym.py differentiates them symbolically into csrc/hilo_models_sym.h (the
derivative code of the interior-point engine for these models), and the tests build the same models through the public
`Model.set_dynamical_equations` front-end and compare the run-time compiled result with the precompiled one bit for bit."""
from . import expr as _e

FUNCTOR = {'chemostat4': 'Chemostat4', 'pendulum4': 'Pendulum4', 'cstr3': 'Cstr3'}
# models whose expressions are only used to ANALYSE the structure (Hessian sparsity of the general policy): no generated code
STRUCTURE_ONLY = ('robot6',)

# CSTR_Example.ipynb cell 4
CSTR = dict(T_0=400., tau=60., k_A=5000., k_B=1e6, E_A=1e4, E_B=1.5e4, R=1.987, dH=-5000., rho=1., Cp=1000., C_A_0=1., V=100.)


def cstr_equations(x, u, lib=None):
    """Right-hand side and reaction rate exactly as the notebook writes them (cell 6), on any symbol type."""
    lib = _e if lib is None else lib
    c = CSTR
    C_A, C_B, T, Q = x[0], x[1], x[2], u[0]
    r = c['k_A'] * lib.exp((-c['E_A']) / (c['R'] * T)) * C_A - c['k_B'] * lib.exp((-c['E_B']) / (c['R'] * T)) * C_B
    dC_A = 1 / c['tau'] * (c['C_A_0'] - C_A) - r
    dC_B = -1 / c['tau'] * C_B + r
    dT = -(c['dH'] * r) / (c['rho'] * c['Cp']) + 1 / c['tau'] * (c['T_0'] - T) + Q / (c['rho'] * c['Cp'] * c['V'])
    return [dC_A, dC_B, dT], r


def define(m, name):
    """Declares states, inputs, parameters and equations of zoo model `name` on the symbolic model `m`."""
    if name == 'chemostat4':
        x = m.set_dynamical_states(['X', 'S', 'P', 'I'])
        u = m.set_inputs(['DS', 'DI'])
        p = m.set_parameters(['Sf', 'If', 'ISF', 'IRF'])
        X, S, Pr, I = x
        phi = 0.407 * S / (0.108 + S + S * S / 14814.0)
        mu = phi * (p[2] + 0.22 * p[3] / (0.22 + I))
        Rs = 2.0 * mu
        Rfp = phi * (0.0005 + I) / (0.022 + I)
        D = u[0] + u[1]
        m.set_dynamical_equations([mu * X - D * X, -(Rs * X) - D * S + u[0] * p[0], Rfp * X - D * Pr, -(D * I) + u[1] * p[1]])
        m.set_measurement_equations([X, Pr])
    elif name == 'pendulum4':
        x = m.set_dynamical_states(['x', 'v', 'theta', 'omega'])
        u = m.set_inputs(['F'])
        M, mm, l, g = 5.0, 1.0, 1.0, 9.81
        s, c = _e.sin(x[2]), _e.cos(x[2])
        dv = 1.0 / (M + mm - mm * c) * (mm * g * s - mm * l * s * x[3] * x[3] + u[0])
        m.set_dynamical_equations([x[1], dv, x[3], 1.0 / l * (dv * c + g * s)])
        m.set_measurement_equations([x[0], x[1], x[2], x[3]])
    elif name == 'cstr3':
        x = m.set_dynamical_states(['C_A', 'C_B', 'T'])
        u = m.set_inputs(['Q'])
        m.set_dynamical_equations(cstr_equations(x, u)[0])
        m.set_measurement_equations([cstr_equations(x, u)[1]])
    elif name == 'robot6':
        x = m.set_dynamical_states(['px', 'vx', 'py', 'vy', 'psi', 'omega'])
        u = m.set_inputs(['a', 'alpha'])
        m.set_dynamical_equations([x[1], u[0] * _e.cos(x[4]), x[3], u[0] * _e.sin(x[4]), x[5], 1.0 * u[1]])
        m.set_measurement_equations([x[0], x[2]])
    else:
        raise ValueError(name)
    return m
