"""Structural Hessian sparsity of a shooting interval (hilo_mpc_amd/sparsity.py) against numbers: the exact second derivatives of
the Runge-Kutta map from the oracle (oracle/shooting.py) must vanish wherever the pattern says zero, for the zoo models that have
an expression form; the pattern of BASELINE configuration 5 is the expected one."""
import numpy as np
import pytest

from hilo_mpc_amd import Model, zoo_expr
from hilo_mpc_amd.expr import hessian_structure, sin
from hilo_mpc_amd.sparsity import stage_hessian_pattern


@pytest.mark.parametrize('name', ['robot6', 'chemostat4', 'pendulum4', 'cstr3'])
def test_pattern_covers_the_exact_hessian_of_the_runge_kutta_map(name):
    from oracle import models
    from oracle.shooting import ShootingMap
    m = zoo_expr.define(Model(name=name + '_s'), name)
    om = models.get(name)
    nx, nu = om.nx, om.nu
    P = stage_hessian_pattern(m._ode, nx, nu)
    assert P.shape == (nx + nu, nx + nu) and np.array_equal(P, P.T) and np.all(np.diag(P) == 1)
    sm = ShootingMap(om, 4)
    rng = np.random.default_rng(0)
    base = {'robot6': [0., 1.5, 0., 1.3, .6, .2], 'chemostat4': [.1, 40., .5, .2], 'pendulum4': [0., .1, .3, -.2],
            'cstr3': [.5, .5, 430.]}[name]
    for _ in range(4):
        x = np.array(base) * (1 + .2 * rng.uniform(-1, 1, nx)) + .05 * rng.normal(size=nx)
        u = rng.uniform(.1, .5, (1, nu)) * (1e4 if name == 'cstr3' else 1.)
        p = np.array([[100., 4., 1., .3]]) if name == 'chemostat4' else np.zeros((1, 0))
        _, _, H = sm(x[None], u, p, .1)
        Hn = np.abs(H[0]).max(axis=0)                       # [nz, nz]: largest entry over the state components
        scale = max(Hn.max(), 1e-300)
        assert np.all(Hn[P == 0] <= 1e-14 * scale), (name, Hn, P)
    if name == 'robot6':
        assert int(np.triu(P, 1).sum()) == 6                # {psi, omega, a, alpha} couple; the translational states do not


def test_configuration_5_pattern():
    """Path following on the mobile robot with a soft speed limit (BASELINE configuration 5): 8 pair directions of 45."""
    m = zoo_expr.define(Model(name='r6'), 'robot6')
    nx, nu = 6, 2
    theta = __import__('hilo_mpc_amd').expr.Expr('theta', value=0, name='theta')
    Wz = np.zeros((10, 10))
    Wz[7, 7] = Wz[8, 8] = .1
    vx, vy = m.x['vx'], m.x['vy']
    P = stage_hessian_pattern(m._ode, nx, nu, nth=1, Wz=Wz, exprs=[vx ** 2 + vy ** 2],
                              path_terms=[(0, sin(theta)), (2, sin(2 * theta))], path_weights=np.diag([10., 10.]))
    want = {(4, 5), (4, 7), (4, 8), (5, 7), (5, 8), (7, 8), (0, 6), (2, 6)}
    got = {(a, b) for a in range(10) for b in range(a + 1, 10) if P[a, b]}
    assert got == want
    # the continuous objective evaluates the cost along the map: its pairs are lifted through what reaches px, py and theta
    Pc = stage_hessian_pattern(m._ode, nx, nu, nth=1, Wz=Wz, exprs=[vx ** 2 + vy ** 2],
                               path_terms=[(0, sin(theta)), (2, sin(2 * theta))], path_weights=np.diag([10., 10.]), composed=True)
    assert np.all(Pc >= P) and Pc[1, 9] == 1 and Pc[7, 9] == 1          # (vx, u_theta), (a, u_theta) through px <- vx <- a
    # coupled path terms (off-diagonal weight): px and py couple as well
    Pw = stage_hessian_pattern(m._ode, nx, nu, nth=1, path_terms=[(0, sin(theta)), (2, sin(2 * theta))],
                               path_weights=np.array([[10., 1.], [1., 10.]]))
    assert Pw[0, 2] == 1 and P[0, 2] == 0


def test_hessian_structure_rules():
    m = Model(name='t')
    x = m.set_dynamical_states(['a', 'b', 'c'])
    dep, prs = hessian_structure(x[0] * x[1] + 3. * x[2])
    assert dep == {('x', 0), ('x', 1), ('x', 2)} and prs == {(('x', 0), ('x', 1))}
    _, prs = hessian_structure(sin(x[0] + x[2]) / x[1])
    assert (('x', 0), ('x', 2)) in prs and (('x', 1), ('x', 1)) in prs and (('x', 0), ('x', 1)) in prs
    assert hessian_structure(2. * x[0] - x[1])[1] == set()


def test_quadratic_cost_along_the_map_couples_what_reaches_the_weighted_state():
    """Continuous objective: the weight on a state acts at the stage points of the map - (P, P) curvature becomes curvature between
    every pair of variables that reach P."""
    m = zoo_expr.define(Model(name='c4'), 'chemostat4')
    Wz = np.zeros((6, 6))
    Wz[2, 2] = 10.
    P0 = stage_hessian_pattern(m._ode, 4, 2, Wz=Wz)
    Pc = stage_hessian_pattern(m._ode, 4, 2, Wz=Wz, composed=True)
    assert np.all(Pc >= P0) and Pc[0, 1] == 1                    # (X, S): both feed the product balance
    lin = Model(name='lin')
    x = lin.set_dynamical_states(['a', 'b'])
    u = lin.set_inputs(['v'])
    lin.set_dynamical_equations([x[1], u[0]])
    W = np.zeros((3, 3))
    W[0, 0] = 1.
    assert int(np.triu(stage_hessian_pattern(lin._ode, 2, 1, Wz=W), 1).sum()) == 0
    assert int(np.triu(stage_hessian_pattern(lin._ode, 2, 1, Wz=W, composed=True), 1).sum()) == 3   # a <- b <- v
