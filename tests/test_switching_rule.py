"""CPU: the decision rule the interior-point engine uses for the switching condition of the filter line search (W&B eq. 19,
csrc/hilo_ocp.h::Ocp::switching):  alpha (-dphi)^s_phi > delta_ls theta^s_theta  is decided in the logarithm with a SINGLE-precision
log2 of the mantissas whenever the two sides are more than 2^(1e-4) apart, and by the pow() expression otherwise (also for arguments so extreme that a power over- or underflows).  Restated here in
numpy (float32 log2 of frexp's mantissa + the exponent, the sums in double) and compared with the exact expression: the same
decision for every input, and the fallback is taken only in a band of relative width ~1e-4 around equality."""
import numpy as np

S_PHI, S_THETA, DELTA = 2.3, 1.1, 1.0          # IPOPT's defaults (W&B sec. 2.3)


def _lg2(x):
    m, e = np.frexp(x)
    return e.astype(np.float64) + np.log2(m.astype(np.float32)).astype(np.float64)


def _rule(alpha, nd, th0, dls=DELTA):
    with np.errstate(over='ignore', under='ignore', invalid='ignore'):
        exact = alpha * np.power(nd, S_PHI) > dls * np.power(th0, S_THETA)
    plain = (alpha > 1e-30) & (alpha < 1e30) & (nd > 1e-100) & (nd < 1e100) & (th0 > 1e-100) & (th0 < 1e100) & (dls > 1e-30) & (dls < 1e30)
    with np.errstate(divide='ignore', invalid='ignore'):
        d = _lg2(np.where(plain, alpha, 1.)) + S_PHI * _lg2(np.where(plain, nd, 1.)) - _lg2(np.full_like(alpha, dls)) \
            - S_THETA * _lg2(np.where(plain, th0, 1.))
    fast = plain & (np.abs(d) > 1e-4)
    return np.where(fast, d > 0., exact), exact, fast


def test_same_decision_as_the_pow_expression_everywhere():
    rng = np.random.default_rng(0)
    n = 400000
    alpha = 2. ** rng.uniform(-40, 0, n)
    nd = 10. ** rng.uniform(-300, 300, n)
    th0 = 10. ** rng.uniform(-300, 2, n)
    got, exact, fast = _rule(alpha, nd, th0)
    assert np.array_equal(got, exact) and fast.mean() > .05          # (most of this cube is outside the fast path's ranges)
    alpha, nd, th0 = 2. ** rng.uniform(-40, 0, n), 10. ** rng.uniform(-60, 20, n), 10. ** rng.uniform(-60, 2, n)
    got, exact, fast = _rule(alpha, nd, th0)
    assert np.array_equal(got, exact) and fast.mean() > .999
    # on the boundary and a hair on either side of it: theta chosen so that the two sides agree to a relative 1e-9 .. 1e-3
    nd = 10. ** rng.uniform(-8, 2, n)
    alpha = 2. ** rng.integers(-12, 1, n).astype(float)
    th_eq = (alpha * nd ** S_PHI / DELTA) ** (1. / S_THETA)
    for rel in (0., 1e-9, 1e-6, 3e-5, 1e-4, 1e-3):
        for sgn in (-1., 1.):
            got, exact, fast = _rule(alpha, nd, th_eq * (1. + sgn * rel))
            assert np.array_equal(got, exact)
            if rel >= 1e-3:
                assert fast.all()                  # far enough from equality: never the slow path
            if rel <= 1e-6:
                assert not fast.any()              # inside the band: always the pow() expression


def test_degenerate_arguments_take_the_pow_expression():
    a = np.array([1., 1., 1., 0., np.inf, 1.])
    nd = np.array([0., 1., np.inf, 1., 1., np.nan])
    th = np.array([1., 0., 1., 1., 1., 1.])
    with np.errstate(all='ignore'):
        got, exact, fast = _rule(a, nd, th)
    assert not fast.any() and np.array_equal(got, exact)
