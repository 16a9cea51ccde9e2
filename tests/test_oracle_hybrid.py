"""CPU: the numeric statement of a model with a learned term (oracle/models.py::NumericHybridModel - the oracle side of
tests/test_kf_learned_gpu.py) against the symbolic statement (`chemostat4_gp`: the kernel sum written out, Runge-Kutta map and
Jacobian by sympy) on a training set small enough for sympy, and the filter step built on either."""
import numpy as np

from oracle import kf as okf, models as M


def _pair(n=7, seed=0):
    rng = np.random.default_rng(seed)
    Xt = np.stack([rng.uniform(0, 40, n), rng.uniform(0, 4, n)])
    al = rng.normal(size=n)
    sym = M.chemostat4_gp(Xt, al, [8., 1.5], .7)
    num = M.NumericHybridModel(M.chemostat4_mu(), M.se_mean_term(Xt, al, [8., 1.5], .7, [1, 3]))
    return sym, num, rng


def _points(rng, B=6):
    x = np.array([.1, 30., .5, .4]) * (1 + .2 * rng.uniform(-1, 1, (B, 4)))
    u = rng.uniform(0, .3, (B, 2))
    p = np.tile([100., 4., 1., 0.], (B, 1))
    return x, u, p


def test_right_hand_side_map_and_jacobians_equal_the_symbolic_statement():
    sym, num, rng = _pair()
    x, u, p = _points(rng)
    np.testing.assert_allclose(num.f(x, u, p, 1.), sym.f(x, u, p, 1.), rtol=1e-13, atol=1e-14)
    np.testing.assert_allclose(num.fx(x, u, p, 1.), sym.fx(x, u, p, 1.), rtol=1e-12, atol=1e-14)
    for order in (1, 2, 3, 4):
        sd, nd = sym.discretize(order), num.discretize(order)
        assert nd.discrete and nd.nx == 4 and nd.ny == 2
        np.testing.assert_allclose(nd.f(x, u, p, .7), sd.f(x, u, p, .7), rtol=1e-13, atol=1e-14)
        np.testing.assert_allclose(nd.fx(x, u, p, .7), sd.fx(x, u, p, .7), rtol=1e-11, atol=1e-13)
    np.testing.assert_array_equal(num.h(x, u, p, 1.), sym.h(x, u, p, 1.))
    np.testing.assert_array_equal(num.hx(x, u, p, 1.), sym.hx(x, u, p, 1.))
    # the gradient of the learned term against central differences
    term = num.term
    v, g = term(x)
    for i in range(4):
        e = np.zeros(4)
        e[i] = 1e-6
        np.testing.assert_allclose((term(x + e)[0] - term(x - e)[0]) / 2e-6, g[:, i], rtol=1e-6, atol=1e-9)


def test_filter_steps_on_the_numeric_model_equal_those_on_the_symbolic_one():
    sym, num, rng = _pair(seed=1)
    x, u, p = _points(rng, 5)
    A = rng.normal(size=(5, 4, 4))
    P = .1 * A @ np.swapaxes(A, 1, 2) + .5 * np.eye(4)
    y = x[:, [0, 2]] + .01 * rng.normal(size=(5, 2))
    a, ya = okf.kf_step(sym.discretize(4), okf.pack(x, P), y, u, p, 1e-4, 1e-2, 1.)
    b, yb = okf.kf_step(num.discretize(4), okf.pack(x, P), y, u, p, 1e-4, 1e-2, 1.)
    np.testing.assert_allclose(b, a, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(yb, ya, rtol=1e-13)
    a, _ = okf.ukf_step(sym.discretize(4), okf.pack(x, P), y, u, p, 1e-4, 1e-2, 1., alpha=1.)
    b, _ = okf.ukf_step(num.discretize(4), okf.pack(x, P), y, u, p, 1e-4, 1e-2, 1., alpha=1.)
    np.testing.assert_allclose(b, a, rtol=1e-10, atol=1e-12)
    # continuous-time filter (kf.py:97-110): [x; vec P] integrated with the numeric right-hand side and Jacobian
    a, _ = okf.kf_step(sym, okf.pack(x[:2], P[:2]), y[:2], u[:2], p[:2], 1e-4, 1e-2, .5)
    b, _ = okf.kf_step(num, okf.pack(x[:2], P[:2]), y[:2], u[:2], p[:2], 1e-4, 1e-2, .5)
    np.testing.assert_allclose(b, a, rtol=1e-8, atol=1e-10)
