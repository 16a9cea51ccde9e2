"""CPU: the general NMPC oracle (path following, stage constraints; oracle/nmpc_gen.py) is 'parity unpinned'.  It is
checked here (a) against the plain oracle when no extra feature is active (identical iterates), (b) by an independent
scipy SLSQP solve of the reference's transcription with inequality rows and the shared slack, (c) for the integer
bookkeeping of the decision vector.  Also: the host-side expression compiler against a Python model of the device VM."""
import math

import numpy as np
import pytest
from scipy.optimize import minimize

from oracle.nmpc import DenseIpm, IpmOptions
from oracle.nmpc_gen import GenIpm
from tests.problems import C2, C2H, C5, C5S, c2_x0, c5_x0, oracle_gen, oracle_problem


def test_general_oracle_reduces_to_the_plain_one():
    x0 = c2_x0(3)
    r = GenIpm(oracle_gen(C2)).solve(x0, C2['p'])
    r0 = DenseIpm(oracle_problem(C2)).solve(x0, C2['p'])
    assert np.array_equal(r['iters'], r0['iters']) and np.array_equal(r['status'], r0['status'])
    np.testing.assert_array_equal(r['f'], r0['f'])


def test_bookkeeping_c5():
    pb = oracle_gen(C5)
    # theta-augmented nx = 7, nu = 3; + one shared slack; 2 constraint rows per stage in g (soft)
    assert (pb.nxa, pb.nua, pb.ne) == (7, 3, 1)
    assert pb.n_v == 51 * 7 + 50 * 3 + 1 and pb.n_g == 50 * (7 + 2)
    assert pb.x_ind[1] == list(range(7, 14)) and pb.u_ind[0] == list(range(357, 360)) and pb.e_ind == [507]
    assert pb.nrow == 1 and pb.row_ref == [0]            # -c - e <= +inf constrains nothing: dropped
    assert pb.u_guess[-1] == pytest.approx(2e-4) and pb.u_lb[-1] == 1e-4 and pb.x_lb[-1] == 0.


def _slsqp(pb, ipm, res, b, p, strict=True):
    """The reference's NLP in its own variables [xa_0 (theta_0 only) .. | ua | e] with x_0 substituted."""
    N, nxa, nua, nx = pb.N, pb.nxa, pb.nua, pb.nx
    x0s = res['x0'][b]
    v_ipm = ipm.to_v(res)[b]
    free = np.ones(pb.n_v, dtype=bool)
    free[:nx] = False

    def full(w):
        v = np.empty(pb.n_v)
        v[:nx] = x0s
        v[free] = w
        return v

    def split(v):
        X = v[:(N + 1) * nxa].reshape(N + 1, nxa)
        U = v[(N + 1) * nxa:(N + 1) * nxa + N * nua].reshape(N, nua)
        return X, U, v[(N + 1) * nxa + N * nua:]

    def obj(w):
        X, U, E = split(full(w))
        J = 0.
        for k in range(N):
            zk = np.concatenate([X[k], U[k]])[None]
            z = zk - pb.zrefa
            J += (z @ pb.Wza @ z.T).item() + pb._lp[0](zk)[0] + (float(E @ pb.We @ E) if pb.ne else 0.)
        d = X[N][None] - pb.xrefNa
        return J + (d @ pb.WNa @ d.T).item() + pb._Vp[0](X[N][None])[0]

    def eq(w):
        X, U, E = split(full(w))
        return np.concatenate([X[k + 1] - pb.phia(X[k][None], U[k][None], np.atleast_2d(p))[0] for k in range(N)])

    def ineq(w):                                             # >= 0
        X, U, E = split(full(w))
        out = []
        for k in range(N if pb.nrow else 0):
            d = pb._d(np.concatenate([X[k], U[k]])[None], E[None])[0]
            out += [np.where(np.isfinite(pb.dub), pb.dub - d, 1.), np.where(np.isfinite(pb.dlb), d - pb.dlb, 1.)]
        if pb.nt:                                            # terminal rows on the integrated end state (mpc.py:1693-1700)
            ct = pb._ct(pb.phia(X[N - 1][None], U[N - 1][None], np.atleast_2d(p))[:, :pb.nx] * pb.sx)[0]
            out += [np.where(np.isfinite(pb.tub), pb.tub - ct, 1.), np.where(np.isfinite(pb.tlb), ct - pb.tlb, 1.)]
        return np.concatenate(out)
    lb = np.concatenate([np.tile(pb.x_lb, N + 1), np.tile(pb.u_lb, N), np.zeros(pb.ne)])[free]
    ub = np.concatenate([np.tile(pb.x_ub, N + 1), np.tile(pb.u_ub, N), pb.e_ub if pb.ne else np.zeros(0)])[free]
    w0 = np.clip(v_ipm[free] + 1e-3 * np.random.default_rng(b).normal(size=free.sum()), lb, ub)
    cons = [{'type': 'eq', 'fun': eq}] + ([{'type': 'ineq', 'fun': ineq}] if pb.nrow or pb.nt else [])
    sol = minimize(obj, w0, method='SLSQP', bounds=list(zip(lb, ub)), constraints=cons, options={'ftol': 1e-11, 'maxiter': 800})
    # strict=False: with finite-difference gradients SLSQP may stop at its line search next to the optimum (status 8) - the
    # point it reached is compared all the same, and the interior point's cost must not be worse
    assert sol.success or (not strict and sol.status == 8), sol.message
    np.testing.assert_allclose(sol.fun, obj(v_ipm[free]), rtol=1e-6, atol=1e-8)
    assert strict or obj(v_ipm[free]) <= sol.fun + 1e-9 * abs(sol.fun)
    np.testing.assert_allclose(sol.x, v_ipm[free], rtol=5e-4 if strict else 5e-3, atol=5e-4 if strict else 5e-3)
    assert np.abs(eq(v_ipm[free])).max() < 1e-8 and ineq(v_ipm[free]).min() > -1e-6   # bounds are relaxed by 1e-8 max(1, |b|) like IPOPT


def test_hard_constraint_vs_slsqp():
    spec = dict(C2H, N=6, constraint=dict(expr=['X * S'], lb=[-np.inf], ub=[20.]))
    pb = oracle_gen(spec)
    ipm = GenIpm(pb)
    res = ipm.solve(c2_x0(1), spec['p'])
    assert res['status'][0] == 1 and res['kkt'][0] <= 1e-8
    assert (res['X'][0, :-1, 0] * res['X'][0, :-1, 1]).max() > 19.99       # active
    _slsqp(pb, ipm, res, 0, spec['p'])


def test_hard_terminal_constraint_vs_slsqp():
    spec = dict(C2, N=6, terminal_constraint=dict(expr=['X + P', 'S'], lb=[-np.inf, 30.], ub=[1.0, np.inf]))
    pb = oracle_gen(spec)
    ipm = GenIpm(pb)
    res = ipm.solve(c2_x0(1), spec['p'])
    assert res['status'][0] == 1 and res['kkt'][0] <= 1e-8
    xe = res['X'][0, -1] * pb.sx
    assert xe[0] + xe[2] > 1. - 1e-6                                       # X + P <= 1 is active at the end state
    assert pb.n_g == 6 * 4 + 2                                              # two rows after the last defect (mpc.py:1693-1700)
    _slsqp(pb, ipm, res, 0, spec['p'])


def test_free_initial_state_vs_slsqp():
    """fix_x0=False (mpc.py:797-807): x_0 is a bounded variable; independent SLSQP solve of the same transcription."""
    spec = dict(C2, N=5, x_lb=[1., 10., 0., 0.], x_ub=[8., 60., 5., 20.], x_guess=[4., 30., 1., 5.])
    pb = oracle_gen(spec)
    ipm = GenIpm(pb, free_x0=True)
    res = ipm.solve(c2_x0(1), spec['p'])
    assert res['status'][0] == 1 and res['kkt'][0] <= 1e-8
    N, nx, nu = pb.N, pb.nx, pb.nu
    p = np.atleast_2d(spec['p'])

    def split(v):
        return v[:(N + 1) * nx].reshape(N + 1, nx), v[(N + 1) * nx:].reshape(N, nu)

    def obj(v):
        X, U = split(v)
        J = 0.
        for k in range(N):
            z = np.concatenate([X[k], U[k]]) - pb.zrefa[0] if pb.zrefa.ndim > 1 else np.concatenate([X[k], U[k]]) - pb.zrefa
            J += float(z @ pb.Wza @ z)
        d = X[N] - (pb.xrefNa[0] if pb.xrefNa.ndim > 1 else pb.xrefNa)
        return J + float(d @ pb.WNa @ d)

    def eq(v):
        X, U = split(v)
        return np.concatenate([X[k + 1] - pb.phia(X[k][None], U[k][None], p)[0] for k in range(N)])
    v_ipm = ipm.to_v(res)[0]
    lb = np.concatenate([np.tile(pb.x_lb, N + 1), np.tile(pb.u_lb, N)])
    ub = np.concatenate([np.tile(pb.x_ub, N + 1), np.tile(pb.u_ub, N)])
    w0 = np.clip(v_ipm + 1e-3 * np.random.default_rng(0).normal(size=v_ipm.size), lb, ub)
    sol = minimize(obj, w0, method='SLSQP', bounds=list(zip(lb, ub)), constraints=[{'type': 'eq', 'fun': eq}],
                   options={'ftol': 1e-12, 'maxiter': 800})
    assert sol.success, sol.message
    np.testing.assert_allclose(sol.fun, obj(v_ipm), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(sol.x, v_ipm, rtol=5e-4, atol=5e-4)
    assert abs(v_ipm[0] - c2_x0(1)[0, 0] / pb.sx[0]) > 1e-2                  # not the measured state


SOFT_T = dict(expr=['X + P'], lb=[0.5], ub=[1.0], soft=True, weight=[[50.]])


@pytest.mark.parametrize('spec', [
    dict(C2, N=4, terminal_constraint=SOFT_T),                                                        # soft terminal rows
    dict(C2H, N=4, terminal_constraint=dict(expr=['S'], lb=[45.], ub=[np.inf], soft=True)),           # + hard stage row
    dict(C2H, N=4, terminal_constraint=dict(expr=['X + P', 'S'], lb=[-np.inf, 30.], ub=[1., np.inf])),  # hard terminal rows
    dict(C2, N=4, constraint=dict(expr=['X * S'], lb=[5.], ub=[60.], soft=True)),                    # soft stage rows
], ids=['soft_terminal', 'hard_stage_soft_terminal', 'hard_terminal', 'soft_stage'])
def test_derivatives_by_finite_differences(spec):
    """Gradient, constraint Jacobian and Lagrangian Hessian of the oracle against central differences of its own value
    functions (the value path is the plain restatement of mpc.py:1654-1725)."""
    pb = oracle_gen(spec)
    ipm = GenIpm(pb)
    rng = np.random.default_rng(1)
    B = 2
    data = {'x0': c2_x0(B) / pb.sx, 'p': np.broadcast_to(np.atleast_2d(spec['p']), (B, 4))}
    w = np.abs(rng.normal(size=(B, ipm.nw))) + 0.5
    lam = rng.normal(size=(B, ipm.m))
    f, g, c, J, W = ipm.eval_all(w, lam, data)
    f2, c2 = ipm.eval_fc(w, data)
    np.testing.assert_allclose(f, f2, rtol=1e-14)
    np.testing.assert_allclose(c, c2, rtol=1e-14, atol=1e-14)

    def lag_grad(wq):
        _, gq, _, Jq, _ = ipm.eval_all(wq, lam, data)
        return gq + np.einsum('bm,bmn->bn', lam, Jq)
    h = 1e-6
    for i in range(ipm.nw):
        wp, wm = w.copy(), w.copy()
        wp[:, i] += h
        wm[:, i] -= h
        (fp, cp), (fm, cm) = ipm.eval_fc(wp, data), ipm.eval_fc(wm, data)
        np.testing.assert_allclose(g[:, i], (fp - fm) / (2 * h), rtol=2e-6, atol=2e-5 + 1e-9 * np.abs(f).max() / h)   # round-off of f / h
        np.testing.assert_allclose(J[:, :, i], (cp - cm) / (2 * h), rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(W[:, :, i], (lag_grad(wp) - lag_grad(wm)) / (2 * h), rtol=2e-5, atol=2e-4)


def test_soft_terminal_constraint_bookkeeping_and_solution():
    spec = dict(C2, N=6, terminal_constraint=SOFT_T)
    pb = oracle_gen(spec)
    assert pb.n_v == 7 * 4 + 6 * 2 + 1 and pb.eT_ind == [40] and pb.n_g == 6 * 4 + 2     # mpc.py:1540-1548, :1687-1690
    ipm = GenIpm(pb)
    res = ipm.solve(c2_x0(2), spec['p'])
    assert np.all(res['status'] == 1)
    eT = res['w'][:, ipm.o_eT:ipm.o_s][:, 0]
    xm = res['X'][:, -2] * pb.sx                                                          # x_{N-1}: mpc.py:1685 uses x_ii
    np.testing.assert_allclose(xm[:, 0] + xm[:, 2] - eT, 1.0, atol=1e-6)                  # X + P - e_T = ub: active
    assert np.all(eT > 0.1)
    lam = ipm.lam_g(res)
    assert lam.shape[1] == pb.n_g and np.all(np.abs(lam[:, 5 * 4 + 4 + 1]) < 1e-7) and np.all(lam[:, 5 * 4 + 4] > 1.)   # ub row active, lb row not


def test_path_following_with_soft_constraint_vs_slsqp():
    spec = dict(C5S, N=6, constraint=dict(C5S['constraint'], weight=[[10.]]))    # milder penalty: SLSQP-friendly scaling
    pb = oracle_gen(spec)
    ipm = GenIpm(pb)
    x0 = c5_x0(4)
    res = ipm.solve(x0, [])
    assert np.all(res['status'] == 1) and np.all(res['kkt'] <= 1e-8)
    b = int(np.argmax(res['E'][:, 0]))
    assert res['E'][b, 0] > 1e-3                                           # the slack is in use
    _slsqp(pb, ipm, res, b, np.zeros(0))
    lam = ipm.lam_g(res)
    assert lam.shape == (4, pb.n_g)
    assert np.all(lam.reshape(4, pb.N, -1)[:, :, pb.nxa + 1] == 0.)         # the dropped row -c - e <= inf


def test_constraint_on_the_path_variable_vs_slsqp():
    """A stage constraint that involves the path variable (a soft tube around the first path coordinate next to the soft speed
    limit: two slacks) - the independent SLSQP solve of the same transcription agrees."""
    spec = dict(C5S, N=6, constraint=dict(expr=['vx**2 + vy**2', 'px - sin(theta)'], lb=[-np.inf, -0.1], ub=[4., 0.1], soft=True,
                                         weight=[[10., 0.], [0., 10.]]))
    pb = oracle_gen(spec)
    assert pb.ne == 2 and pb.nrow == 3                                   # ub row of the speed limit, both rows of the tube
    ipm = GenIpm(pb)
    res = ipm.solve(c5_x0(4), [])
    assert np.all(res['status'] == 1) and np.all(res['kkt'] <= 1e-8)
    b = int(np.argmax(res['E'][:, 1]))
    assert res['E'][b, 1] > 1e-4                                          # the tube's slack is in use
    _slsqp(pb, ipm, res, b, np.zeros(0))


# ---- expression compiler (host) vs a Python model of the device interpreter -------------------------------------------
def _run(prog, x, u, p):
    n = int(prog[0])
    st = []
    for q in range(1, 1 + n, 2):
        op, a = int(prog[q]), prog[q + 1]
        if op == 0: st.append(a)
        elif op == 1: st.append(x[int(a)])
        elif op == 2: st.append(u[int(a)])
        elif op == 3: st.append(p[int(a)])
        elif op in (10, 11, 12, 13):
            b, c = st.pop(), st.pop()
            st.append({10: c + b, 11: c - b, 12: c * b, 13: c / b}[op])
        else:
            b = st.pop()
            st.append({14: -b, 15: b * b, 16: math.sin(b), 17: math.cos(b), 18: math.exp(b),
                       19: math.log(b) if b > 0 else float('nan'), 20: math.sqrt(abs(b)), 21: b ** int(a)}[op])
        assert len(st) <= 8
    assert len(st) == 1
    return st[0]


def test_expression_compiler():
    from hilo_mpc_amd import Model, expr
    m = Model('robot6')
    vx, vy, psi = m.x['vx'], m.x['vy'], m.x[4]
    a = m.u['a']
    e = (vx ** 2 + vy ** 2) / (1. + expr.cos(psi) ** 2) - 3 * a + expr.sqrt(vx * vx + 1.) ** -1 + expr.exp(-vy) * expr.log(2. + a * a)
    x = [.1, 1.2, -.3, .7, .4, 0.]
    u = [.9, -.2]
    val = (1.2 ** 2 + .7 ** 2) / (1 + math.cos(.4) ** 2) - 3 * .9 + 1 / math.sqrt(1.2 * 1.2 + 1) + math.exp(-.7) * math.log(2 + .81)
    assert _run(e.program(), x, u, []) == pytest.approx(val, rel=1e-14)
    blk = expr.compile_block([vx + 1., 2 * vy])
    assert blk[0] == 6 and blk[7] == 6 and len(blk) == 14
    theta = expr.Expr('theta', value=0, name='theta')
    assert _run(expr.sin(2 * theta).program(theta_index=6), x + [.25], u, []) == pytest.approx(math.sin(.5))
    with pytest.raises(ValueError, match="only appear in path references"):
        (theta + vx).program()
    # a general power of a positive base is composed as exp(b log a)
    assert _run((vx ** 2.5).program(), x, u, []) == pytest.approx(1.2 ** 2.5, rel=1e-14)
    with pytest.raises(KeyError):
        m.x['nope']


def _trapezoid(v, x_ind, u_ind, i=2, dt=.25):
    """The reference's own custom function (tests/test_NMPC.py:524-529): trapezoid integral of one state over the nodes."""
    s = 0
    for k in range(len(x_ind) - 1):
        s += (v[x_ind[k][i]] + v[x_ind[k + 1][i]]) / 2 * dt
    return s


def test_custom_constraint_rows_dense_oracle_vs_slsqp():
    """`set_custom_constraints_function` (optimizer.py:1180-1208; rows at the end of g, mpc.py:1741-1744) in the oracle: DENSE rows
    over the whole decision vector, in IPOPT's slack form.  The interior point and scipy SLSQP on the same transcription find the
    same point; the row is active; its multiplier is the last entry of lam_g and positive (upper bound active)."""
    from scipy.optimize import minimize
    from tests.problems import C2, c2_x0, oracle_gen
    spec = dict(C2, N=5)
    x0 = c2_x0(2)
    free = GenIpm(oracle_gen(spec)).solve(x0, C2['p'])
    X = free['X']
    integ = ((X[:, :-1, 2] + X[:, 1:, 2]) / 2 * .25).sum(1)
    ub = float(integ.min() * .9)
    pb = oracle_gen(dict(spec, custom=dict(fun=_trapezoid, lb=0., ub=ub)))
    assert pb.n_g == spec['N'] * 4 + 1 and pb.n_cus == 1
    ipm = GenIpm(pb)
    ref = ipm.solve(x0, C2['p'])
    assert np.all(ref['status'] == 1)
    X = ref['X']
    np.testing.assert_allclose(((X[:, :-1, 2] + X[:, 1:, 2]) / 2 * .25).sum(1), ub, rtol=1e-7)
    lam = ipm.lam_g(ref)
    assert lam.shape == (2, pb.n_g) and np.all(lam[:, -1] > 1.) and np.all(ref['f'] > free['f'])
    # SLSQP on the same NLP (free variables of the oracle without the slacks), instance 0
    nfree = ipm.o_s
    data = {'x0': x0[:1] / pb.sx, 'p': np.atleast_2d(np.asarray(C2['p'], dtype=float))}

    def ev(wv):
        w = np.concatenate([wv, np.zeros(ipm.nw - nfree)])[None]
        f, c = ipm.eval_fc(w, data)
        return f[0], c[0]
    nd = pb.N * pb.nxa
    cons = [{'type': 'eq', 'fun': lambda wv: ev(wv)[1][:nd]},
            {'type': 'ineq', 'fun': lambda wv: ub - ev(wv)[1][-1]}]         # (the slack is zero in ev: the row's value itself)
    sol = minimize(lambda wv: ev(wv)[0], ref['w'][0, :nfree] * (1 + 1e-3), method='SLSQP', bounds=list(zip(ipm.lb[:nfree], ipm.ub[:nfree])),
                   constraints=cons, options={'ftol': 1e-11, 'maxiter': 800})
    # (with finite-difference gradients SLSQP may stop at its line search next to the optimum, status 8 - like _slsqp above)
    assert sol.success or sol.status == 8, sol.message
    np.testing.assert_allclose(sol.fun, ref['f'][0], rtol=1e-6)
    assert ref['f'][0] <= sol.fun + 1e-9 * abs(sol.fun)
    np.testing.assert_allclose(sol.x, ref['w'][0, :nfree], rtol=5e-3, atol=5e-3)


def test_soft_custom_rows_dense_oracle_vs_slsqp():
    """custom['soft'] (mpc.py:1551-1556, :1731-1740): the slack e_cus is the last entry of v, in [0, max_violation], 1e4 e_cus^2 in the
    objective, rows fun - e_cus <= ub and fun + e_cus >= lb at the end of g.  Against scipy SLSQP on the same statement written
    directly (objective of the un-constrained transcription + 1e4 e^2, one inequality fun - e <= ub); the slack opens, the
    multiplier of the upper row is 2e4 e (stationarity in e), the lower row is inactive."""
    from scipy.optimize import minimize
    from tests.problems import C2, c2_x0, oracle_gen
    spec = dict(C2, N=5)
    x0 = c2_x0(2)
    plain = oracle_gen(spec)
    free = GenIpm(plain).solve(x0, C2['p'])
    X = free['X']
    ub = float(((X[:, :-1, 2] + X[:, 1:, 2]) / 2 * .25).sum(1).min() * .9)
    pb = oracle_gen(dict(spec, custom=dict(fun=_trapezoid, lb=0., ub=ub, soft=True, max_violation=3.)))
    assert pb.n_v == plain.n_v + 1 and pb.n_g == plain.n_g + 2 and pb.n_cus == 2 and pb.eT_ind == [pb.n_v - 1]
    ipm = GenIpm(pb, IpmOptions(tol=1e-10))
    assert ipm.ub[ipm.o_eT] == pytest.approx(3., rel=1e-7) and ipm.lb[ipm.o_eT] == pytest.approx(0., abs=1e-7)
    ref = ipm.solve(x0, C2['p'])
    assert np.all(ref['status'] == 1)
    e = ipm.to_v(ref)[:, -1]
    lam = ipm.lam_g(ref)
    X = ref['X']
    integ = ((X[:, :-1, 2] + X[:, 1:, 2]) / 2 * .25).sum(1)
    assert np.all(e > 1e-4)
    np.testing.assert_allclose(integ - e, ub, rtol=1e-7)
    np.testing.assert_allclose(lam[:, -2], 2e4 * e, rtol=1e-6)
    assert np.all(np.abs(lam[:, -1]) < 1e-8)
    hard = GenIpm(oracle_gen(dict(spec, custom=dict(fun=_trapezoid, lb=0., ub=ub))), IpmOptions(tol=1e-10)).solve(x0, C2['p'])
    assert np.all(ref['f'] < hard['f']) and np.all(ref['f'] > free['f'])
    # SLSQP, instance 0: variables [w of the plain transcription | e]
    ip = GenIpm(plain)
    nfree = ip.o_s
    data = {'x0': x0[:1] / plain.sx, 'p': np.atleast_2d(np.asarray(C2['p'], dtype=float))}

    def ev(wv):
        w = np.concatenate([wv[:-1], np.zeros(ip.nw - nfree)])[None]
        f, c = ip.eval_fc(w, data)
        Xw = np.concatenate([data['x0'], wv[ip.o_x:ip.o_u].reshape(plain.N, plain.nxa)], axis=0)
        return f[0] + 1e4 * wv[-1] ** 2, c[0], ((Xw[:-1, 2] + Xw[1:, 2]) / 2 * .25).sum()
    nd = plain.N * plain.nxa
    cons = [{'type': 'eq', 'fun': lambda wv: ev(wv)[1][:nd]},
            {'type': 'ineq', 'fun': lambda wv: ub - (ev(wv)[2] - wv[-1])}]
    w0 = np.concatenate([ref['w'][0, :nfree] * (1 + 1e-3), [e[0] * 1.1]])
    sol = minimize(lambda wv: ev(wv)[0], w0, method='SLSQP', bounds=list(zip(ip.lb[:nfree], ip.ub[:nfree])) + [(0., 3.)],
                   constraints=cons, options={'ftol': 1e-12, 'maxiter': 800})
    assert sol.success or sol.status == 8, sol.message
    np.testing.assert_allclose(sol.fun, ref['f'][0], rtol=1e-6)
    np.testing.assert_allclose(sol.x[-1], e[0], rtol=2e-2)


def test_custom_constraint_decomposition_into_stage_terms():
    """hilo_mpc_amd/custom.py: the user's function called with symbols, split into single-stage terms - expressions shared between
    stages, per-stage coefficients, constant parts; the decomposition reproduces the function's value at random points, and a
    function that multiplies variables of different stages is refused."""
    from hilo_mpc_amd.custom import decompose
    from hilo_mpc_amd.symdiff import Dag
    from tests.problems import symbolic_model
    m = symbolic_model('chemostat4')
    N, nx, nu = 4, 4, 2
    x_ind = [list(range(k * nx, (k + 1) * nx)) for k in range(N + 1)]
    u_ind = [list(range((N + 1) * nx + k * nu, (N + 1) * nx + (k + 1) * nu)) for k in range(N)]
    n_v = (N + 1) * nx + N * nu

    def fun(v, x_ind, u_ind):
        e = 1.5
        for k in range(len(u_ind)):
            e = e + 3 * v[u_ind[k][0]] ** 2 - v[x_ind[k][0]] * v[u_ind[k][1]] / 2
        return [_trapezoid(v, x_ind, u_ind), e - 2 * v[x_ind[N][1]]]
    psi, coef, const = decompose(fun, x_ind, u_ind, n_v, m)
    assert len(psi) == 4 and coef.shape == (2, N + 1, 4) and np.allclose(const, [0., 1.5])
    np.testing.assert_allclose(coef[0, :, 0], [.125, .25, .25, .25, .125])
    rng = np.random.default_rng(0)
    v = rng.uniform(.5, 2., n_v)
    dag, memo = Dag(), {}
    ids = [dag.from_expr(e, memo) for e in psi]
    val = const.copy()
    for k in range(N + 1):
        xk = v[x_ind[k]]
        uk = v[u_ind[k]] if k < N else np.zeros(nu)
        val += coef[:, k] @ np.array(dag.evaluate(ids, xk, uk, []))
    np.testing.assert_allclose(val, [float(e) for e in fun(list(v), x_ind, u_ind)], rtol=1e-13)
    with pytest.raises(NotImplementedError, match="DIFFERENT stages"):
        decompose(lambda v, xi, ui: v[xi[0][0]] * v[xi[1][0]], x_ind, u_ind, n_v, m)
    with pytest.raises(NotImplementedError, match="neither a state nor an input"):
        decompose(lambda v, xi, ui: v[n_v + 1] + v[0], x_ind, u_ind, n_v + 3, m)


def test_custom_constraint_functions_that_do_not_fit_the_stage_structure_are_refused_with_the_reason():
    """hilo_mpc_amd/custom.py::decompose: what is offloaded is a sum over the stages of single-stage terms - anything else says why."""
    import pytest
    from hilo_mpc_amd.custom import MAX_PSI, MAX_ROWS, decompose
    from tests.problems import symbolic_model
    m = symbolic_model('chemostat4')
    N, nx, nu = 4, 4, 2
    x_ind = [list(range(k * nx, (k + 1) * nx)) for k in range(N + 1)]
    u_ind = [list(range((N + 1) * nx + k * nu, (N + 1) * nx + (k + 1) * nu)) for k in range(N)]
    n_v = (N + 1) * nx + N * nu + 1                                       # one trailing entry that is no node variable (a slack)
    psi, coef, const = decompose(lambda v, xi, ui: sum(v[xi[k][2]] for k in range(N + 1)) + 3., x_ind, u_ind, n_v, m)
    assert len(psi) == 1 and coef.shape == (1, N + 1, 1) and np.all(coef == 1.) and const[0] == 3.
    with pytest.raises(NotImplementedError, match='DIFFERENT stages'):
        decompose(lambda v, xi, ui: v[xi[0][0]] * v[xi[1][0]], x_ind, u_ind, n_v, m)
    with pytest.raises(NotImplementedError, match='neither a state nor an input'):
        decompose(lambda v, xi, ui: v[n_v - 1] + v[xi[1][0]], x_ind, u_ind, n_v, m)
    with pytest.raises(NotImplementedError, match='distinct stage expressions'):
        decompose(lambda v, xi, ui: sum(v[xi[1][0]] ** (j + 2) for j in range(MAX_PSI + 1)), x_ind, u_ind, n_v, m)
    with pytest.raises(NotImplementedError, match='custom constraint rows'):
        decompose(lambda v, xi, ui: [v[xi[1][0]]] * (MAX_ROWS + 1), x_ind, u_ind, n_v, m)
    with pytest.raises(ValueError, match='bound'):
        decompose(lambda v, xi, ui: [v[xi[1][0]], v[xi[2][0]]], x_ind, u_ind, n_v, m, n_rows=1)
    with pytest.raises(NotImplementedError, match='decision vector v only'):
        decompose(lambda v, xi, ui: v[xi[1][0]] + m.x[0], x_ind, u_ind, n_v, m)
    # the same stage expression with different factors is ONE psi; products inside a stage are fine
    psi, coef, _ = decompose(lambda v, xi, ui: sum((k + 1.) * v[xi[k][0]] * v[ui[k][1]] for k in range(N)), x_ind, u_ind, n_v, m)
    assert len(psi) == 1 and np.allclose(coef[0, :N, 0], np.arange(1., N + 1)) and coef[0, N, 0] == 0.
