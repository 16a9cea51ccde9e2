"""The C++/OpenMP CPU baseline (oracle/cpu/nmpc_cpu.cpp: Riccati interior point) against the numpy
oracle it restates - same statuses, same iteration counts, same solution - before bench.py times it.  The baseline is test
infrastructure: it is not the product and the product never loads it."""
import numpy as np
import pytest

from oracle import models
from oracle.cpu import CpuNmpc, max_threads
from oracle.nmpc import DenseIpm, NmpcProblem
from tests.problems import C2, c2_x0, oracle_problem


def _compare(pb, x0, p, steps=2, **opt):
    ipm, cpu = DenseIpm(pb, **({'options': opt.pop('options')} if 'options' in opt else {})), CpuNmpc(pb, **opt)
    w, v = None, None
    for k in range(steps):
        ref = ipm.solve(x0, p, w0=w)
        res = cpu.solve(x0, p, v0=v, n_threads=2)
        vr = ipm.to_v(ref)
        assert np.array_equal(res['status'], ref['status'])
        assert np.all(np.abs(res['iters'] - ref['iters']) <= 1), (res['iters'], ref['iters'])   # same algorithm, same path
        ok = ref['status'] == 1
        assert np.max((np.abs(res['v'] - vr) / np.maximum(1., np.abs(vr)))[ok]) < 1e-9
        np.testing.assert_allclose(res['f'][ok], ref['f'][ok], rtol=1e-10)
        np.testing.assert_allclose(res['u0'][ok], ref['u0'][ok], rtol=1e-8, atol=1e-10)
        assert np.all(res['kkt'][ok] <= 1e-8)
        xn = pb.phi(np.atleast_2d(x0) / pb.sx, ref['U'][:, 0], p) * pb.sx
        np.testing.assert_allclose(cpu.plant_step(x0, ref['u0'], p), xn, rtol=1e-12, atol=1e-13)
        x0, w, v = xn, ref['w'], vr                                                             # closed loop, warm-started


def test_c2_cold_and_warm_vs_oracle():
    _compare(oracle_problem(C2), c2_x0(4), C2['p'])


def test_scaled_variables_vs_oracle():
    _compare(oracle_problem(dict(C2, x_scaling=[.1, 40., 2., 1.], u_scaling=[2., 2.])), c2_x0(3), C2['p'])


@pytest.mark.parametrize('order', [1, 2, 3])
def test_lower_order_runge_kutta_vs_oracle(order):
    _compare(oracle_problem(dict(C2, order=order, N=8)), c2_x0(2), C2['p'], steps=1)


def test_pendulum_nonconvex_vs_oracle():
    pb = NmpcProblem(models.get('pendulum4'), dt=.1, N=10, order=4,
                     stage_states=[([1, 2], [10., 5.], [0., 0.])], stage_inputs=[([0], [.1], None)],
                     x_lb=[-5, -10, -10, -10], x_ub=[5, 10, 10, 10], u_lb=[-20.], u_ub=[20.], x_guess=[2.5, 0., .1, 0.], u_guess=[0.])
    _compare(pb, np.array([[2.5, 0., .1, 0.], [1., .5, -.2, .1]]), np.zeros((1, 0)))


def test_thread_count_does_not_change_results():
    pb = oracle_problem(C2)
    cpu = CpuNmpc(pb)
    x0 = c2_x0(12)
    a, b = cpu.solve(x0, C2['p'], n_threads=1), cpu.solve(x0, C2['p'], n_threads=max(2, max_threads()))
    assert np.array_equal(a['v'], b['v']) and np.array_equal(a['iters'], b['iters'])


def test_out_of_scope_descriptor_is_refused():
    with pytest.raises(NotImplementedError):
        CpuNmpc(oracle_problem(dict(C2, Nc=3)))
    with pytest.raises(RuntimeError, match="and pendulum4 only"):
        CpuNmpc(NmpcProblem(models.get('cstr3'), dt=1., N=4))


# ---- the other legs of the baseline: learned growth rate (C4), Kalman filters (C3), the LMPC's QP (C1) --------------------------
def test_c4_learned_growth_rate_vs_oracle():
    """chemostat4 with the GP posterior mean as growth rate (oracle/models.py::chemostat4_gp): the C++ model sums the same 200
    kernel terms in second-order forward mode; same iteration path as the numpy oracle."""
    from oracle.cpu import set_gp
    from tests.problems import C4, C4_GP, c4_training_data, oracle_c4
    pb, post = oracle_c4(dict(C4, N=6))
    X, _ = c4_training_data()
    set_gp(X, post.alpha, C4_GP['length_scales'], C4_GP['signal_variance'])
    _compare(pb, c2_x0(2), C2['p'], steps=2)


@pytest.mark.parametrize('kind', ['ekf', 'ukf'])
def test_filter_steps_vs_oracle(kind):
    from oracle import kf as okf
    from oracle.cpu import kf_steps
    rng = np.random.default_rng(4)
    B, K = 16, 3
    x = np.array([.1, 40., .5, .2]) * (1 + .1 * rng.uniform(-1, 1, (B, 4)))
    P = np.tile(np.eye(4), (B, 1, 1)) * rng.uniform(.5, 1.5, (B, 1, 1))
    u, p = rng.uniform(0, .3, (B, 2)), np.tile([100., 4., 1., 0.], (B, 1))
    y = x[None, :, [0, 2]] * (1 + .02 * rng.normal(size=(K, B, 2)))
    mdl = models.get('chemostat4').discretize(4)
    ref = okf.pack(x, P)
    stepf = okf.kf_step if kind == 'ekf' else okf.ukf_step
    for k in range(K):
        ref, _ = stepf(mdl, ref, y[k], u, p, 1e-4, 1e-2, 1.)
    got = kf_steps(kind, okf.pack(x, P), y, u, p, 1e-4, 1e-2, dt=1., n_threads=2)
    # the unscented transform cancels six digits in its weighted sums (alpha = 1e-3): compare at that level
    np.testing.assert_allclose(got, ref, rtol=1e-9 if kind == 'ekf' else 1e-5, atol=1e-11 if kind == 'ekf' else 1e-8)
    one = kf_steps(kind, okf.pack(x, P), y, u, p, 1e-4, 1e-2, dt=1., n_threads=1)
    assert np.array_equal(one, got)


@pytest.mark.parametrize('Q', [None, [[2., .5], [.5, 1.]]])
def test_lmpc_qp_vs_oracle(Q):
    """diagonal weights (H + Sigma diagonal: one square root per entry, like the device kernel) and a weight with off-diagonal
    entries (the general factorisation)."""
    from oracle.cpu import qp_solve
    from oracle.lmpc import LmpcProblem, solve_qp
    from tests.test_oracle_lmpc import C1
    pb = LmpcProblem(**dict(C1, **({} if Q is None else {'Q': np.array(Q)})), kron_bug=False)
    rng = np.random.default_rng(6)
    xs = rng.uniform(-4, 4, (40, 2))
    bnd = [pb.bounds_for(x0) for x0 in xs]
    lb, ub = np.array([b[0] for b in bnd]), np.array([b[1] for b in bnd])
    res = qp_solve(pb.H, pb.g, pb.Aeq, pb.beq, lb, ub, n_threads=2)
    ref = [solve_qp(pb.H, pb.g, pb.Aeq, pb.beq, l, u) for l, u in zip(lb, ub)]
    st = np.array([r['status'] for r in ref])
    assert np.array_equal(res['status'], st) and set(np.unique(st)) == {1, 3}
    assert np.max(np.abs(res['iters'] - np.array([r['iters'] for r in ref]))) <= 1
    ok = st == 1
    np.testing.assert_allclose(res['x'][ok], np.array([r['x'] for r in ref])[ok], atol=1e-7)     # (the oracle polishes the vertex)


def test_mhe_window_loop_vs_oracle():
    """oracle/cpu/mhe_cpu.cpp against oracle/mhe.py::MheIpm on the C3 window (boxed noise: unique minimiser): cold estimate, then
    the benchmark's loop - a new sample shifts the window, arrival guess = x_2 of the previous solution, warm start = the previous
    solution.  Same statuses, same iteration counts, same solution."""
    from oracle.cpu import CpuMhe
    from oracle.mhe import MheIpm
    from tests.problems import C3B, c3_data, oracle_mhe
    spec = dict(C3B, N=12)
    pb = oracle_mhe(spec)
    ipm, cpu = MheIpm(pb), CpuMhe(pb)
    xa, u, y, _ = c3_data(3, N=12, seed=4)
    rng = np.random.default_rng(8)
    w, v, npar, nx = None, None, pb.np_, pb.nx
    for k in range(3):
        ref = ipm.solve(xa, spec['p'], u, y, w0=w)
        res = cpu.solve(xa, spec['p'], u, y, v0=v, n_threads=2)
        assert np.array_equal(res['status'], ref['status']) and np.all(ref['status'] == 1)
        assert np.all(np.abs(res['iters'] - ref['iters']) <= 1), (res['iters'], ref['iters'])
        assert np.max(np.abs(res['v'] - ref['v']) / np.maximum(1., np.abs(ref['v']))) < 1e-9
        np.testing.assert_allclose(res['f'], ref['f'], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(res['x_opt'], ref['x_opt'], rtol=1e-9, atol=1e-11)
        assert np.all(res['kkt'] <= 1e-8)
        y = np.concatenate([y[:, 1:], (y[:, -1] + .01 * rng.normal(size=(3, 2)))[:, None]], axis=1)
        u = np.concatenate([u[:, 1:], u[:, -1:]], axis=1)
        xa, w, v = ref['v'][:, npar + 2 * nx:npar + 3 * nx] * pb.sx, ref['w'], ref['v']


def test_mhe_scaled_variables_and_unboxed_noise_vs_oracle():
    """Scaling of states, noise and inputs (mhe.py:665-672: costs on un-scaled quantities, u_meas un-divided) and the plain C3
    weights; the unboxed problem is degenerate in the weakly observable states, so the objective is compared there."""
    from oracle.cpu import CpuMhe
    from oracle.mhe import MheIpm
    from tests.problems import C3, C3B, c3_data, oracle_mhe
    xa, u, y, _ = c3_data(2, N=10, seed=6)
    spec = dict(C3B, N=10, x_scaling=[.5, 20., .1, .2], w_scaling=[1e-3, 1e-2, 1e-3, 1e-3], u_scaling=[.1, .05])
    pb = oracle_mhe(spec)
    ref, res = MheIpm(pb).solve(xa, spec['p'], u / pb.su, y), CpuMhe(pb).solve(xa, spec['p'], u / pb.su, y)
    assert np.array_equal(res['status'], ref['status']) and np.all(np.abs(res['iters'] - ref['iters']) <= 1)
    ok = ref['status'] == 1
    assert ok.any() and np.max((np.abs(res['v'] - ref['v']) / np.maximum(1., np.abs(ref['v'])))[ok]) < 1e-8
    pb = oracle_mhe(dict(C3, N=10))
    ref, res = MheIpm(pb).solve(xa, C3['p'], u, y), CpuMhe(pb).solve(xa, C3['p'], u, y)
    assert np.array_equal(res['status'], ref['status']) and np.all(np.abs(res['iters'] - ref['iters']) <= 1)
    np.testing.assert_allclose(res['f'], ref['f'], rtol=1e-8, atol=1e-12)


def test_mhe_out_of_scope_descriptor_is_refused():
    from oracle.cpu import CpuMhe
    from oracle.mhe import MheProblem
    with pytest.raises(RuntimeError, match="chemostat4 only"):
        CpuMhe(MheProblem(models.get('cstr3'), dt=1., N=4))


def test_gp_prediction_vs_oracle():
    """oracle/cpu/gp_cpu.cpp against `Posterior.predict`: mean and variance of the benchmark's posterior (200 points, one length
    scale per feature) at a column count that is no multiple of the block, and a posterior with one length scale, a constant mean
    and the noise-free variance."""
    from oracle import gp as ogp
    from oracle.cpu import gp_predict
    from tests.problems import c4_training_data, oracle_c4
    post = oracle_c4()[1]
    rng = np.random.default_rng(3)
    Xq = np.stack([rng.uniform(0, 40, 1003), rng.uniform(0, 4, 1003)])
    mu, var = post.predict(Xq)
    mc, vc = gp_predict(post, Xq, n_threads=2)
    np.testing.assert_allclose(mc, mu[0], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(vc, var[0], rtol=1e-9, atol=1e-12)
    X, y = c4_training_data()
    post = ogp.Posterior({'type': 'squared_exponential', 'kwargs': dict(length_scales=3., signal_variance=.7)},
                         {'type': 'constant', 'kwargs': {'bias': .2}}, X[:, :50], y[:, :50], 1e-3)
    mu, var = post.predict(Xq[:, :37], noise_free=True)
    mc, vc = gp_predict(post, Xq[:, :37], noise_free=True, n_threads=1)
    np.testing.assert_allclose(mc, mu[0], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(vc, var[0], rtol=1e-9, atol=1e-13)
    with pytest.raises(NotImplementedError):
        gp_predict(ogp.Posterior({'type': 'matern_32', 'kwargs': {}}, {'type': 'zero'}, X[:, :10], y[:, :10], 1e-3), Xq[:, :3])


def test_path_following_with_soft_constraint_vs_oracle():
    """oracle/cpu/pf_cpu.cpp against oracle/nmpc_gen.py::GenIpm on C5 with a short horizon: cold, then two warm-started steps of
    the closed loop.  The C++ solver carries the shared slack as a state (the device engine's form), so the multipliers and the
    iteration path need not be the dense solver's: statuses, minimiser, objective and first input are compared."""
    from oracle.cpu import CpuPathNmpc
    from oracle.nmpc_gen import GenIpm
    from tests.problems import C5S, c5_x0, oracle_gen
    pb = oracle_gen(C5S)
    ipm, cpu = GenIpm(pb), CpuPathNmpc(C5S, pb)
    x0, w_ref, w = c5_x0(3), None, None
    x0[0, [1, 3]] = 1.6, 1.3                  # over the speed limit at k = 0: the slack has to open
    for k in range(3):
        ref = ipm.solve(x0, C5S['p'], w0=w_ref)
        res = cpu.solve(x0, w0=w, n_threads=2)
        vr = ipm.to_v(ref)
        assert np.array_equal(res['status'], ref['status']) and np.all(ref['status'] == 1)
        assert np.all(np.abs(res['iters'] - ref['iters']) <= 3), (res['iters'], ref['iters'])
        assert np.max(np.abs(res['v'] - vr) / np.maximum(1., np.abs(vr))) < 1e-6
        np.testing.assert_allclose(res['f'], ref['f'], rtol=1e-7, atol=1e-10)       # both at tol = 1e-8, penalty weight 1e4
        np.testing.assert_allclose(res['u0'], ref['u0'], rtol=1e-6, atol=1e-7)
        assert np.all(res['kkt'] <= 1e-8)
        if k == 0:
            assert res['v'][0, -1] > .2 and abs(res['v'][0, -1] - ref['E'][0, 0]) < 1e-7
        xn = pb.phi(np.atleast_2d(x0), ref['U'][:, 0, :pb.nu], C5S['p'])
        np.testing.assert_allclose(cpu.plant_step(x0, ref['u0']), xn, rtol=1e-12, atol=1e-13)
        x0, w_ref, w = xn, ref['w'][:, :ipm.o_s], res['w']


def test_path_following_at_the_benchmark_size_vs_oracle():
    """C5 as the benchmark runs it (N = 50, n_v = 508): one cold solve of the C++ leg against the dense numpy solver - same status,
    same iteration count, same point (the dense solve takes ~20 s, the C++ one ~10 ms)."""
    from oracle.cpu import CpuPathNmpc
    from oracle.nmpc_gen import GenIpm
    from tests.problems import C5, c5_x0, oracle_gen
    pb = oracle_gen(C5)
    ipm, cpu = GenIpm(pb), CpuPathNmpc(C5, pb)
    x0 = c5_x0(1)
    ref, res = ipm.solve(x0, C5['p']), cpu.solve(x0, n_threads=1)
    vr = ipm.to_v(ref)
    assert res['status'][0] == ref['status'][0] == 1 and abs(int(res['iters'][0]) - int(ref['iters'][0])) <= 3
    assert vr.shape == res['v'].shape == (1, 508)
    assert np.max(np.abs(res['v'] - vr) / np.maximum(1., np.abs(vr))) < 1e-6
    np.testing.assert_allclose(res['f'], ref['f'], rtol=1e-8)


def test_path_following_out_of_scope_is_refused():
    from oracle.cpu import CpuPathNmpc
    from tests.problems import C2S, C5S, oracle_gen
    with pytest.raises(NotImplementedError):
        CpuPathNmpc(C2S, oracle_gen(C2S))
    spec = dict(C5S, constraint=dict(C5S['constraint'], expr=['vx**2 - vy']))
    with pytest.raises(NotImplementedError):
        CpuPathNmpc(spec, oracle_gen(spec))


def test_configuration_5_on_the_dae_vs_oracle():
    """oracle/cpu/pfdae_cpu.cpp (BASELINE configuration 5 as it is written: the robot's DAE, collocation, soft limit on the algebraic
    state at the collocation points and the node) against the dense oracle oracle/nmpc_coll_gen.py at a short horizon: statuses,
    objective, the [x | u | e] part of the minimiser and the first input; degree 3 (the default) and degree 2; then two warm-started
    steps of the closed loop."""
    from oracle.cpu import CpuPathDaeNmpc
    from oracle.nmpc import IpmOptions
    from oracle.nmpc_coll_gen import GenCollIpm
    from tests.problems import C5DS, c5_x0, oracle_coll_gen
    for degree in (3, 2):
        spec = dict(C5DS, N=6, collocation=dict(degree=degree))
        pb = oracle_coll_gen(spec)
        ipm, cpu = GenCollIpm(pb, IpmOptions(tol=1e-9)), CpuPathDaeNmpc(spec, pb, tol=1e-9)
        x0, w_ref, w = c5_x0(3), None, None
        for k in range(3 if degree == 3 else 1):
            ref = ipm.solve(x0, [], w0=w_ref)
            res = cpu.solve(x0, w0=w, n_threads=2)
            assert np.array_equal(res['status'], ref['status']) and np.all(ref['status'] == 1)
            vx = np.concatenate([ref['X'].reshape(3, -1), ref['U'].reshape(3, -1), ref['E']], axis=1)
            same = np.abs(res['f'] - ref['f']) <= 1e-7 * np.maximum(1., np.abs(ref['f']))     # (non-convex: another minimum is possible)
            assert same.sum() >= 2
            assert np.max(np.abs(res['vx'] - vx)[same] / np.maximum(1., np.abs(vx[same]))) < 5e-6
            np.testing.assert_allclose(res['u0'][same], ref['u0'][same], rtol=5e-6, atol=5e-6)
            assert np.all(res['kkt'] <= 1e-9)
            xn = cpu.plant_step(x0, ref['u0'])
            x0, w_ref, w = xn, ipm.w_from_v(ipm.to_v(ref)), res['w']
    with pytest.raises(NotImplementedError):
        bad = dict(C5DS, constraint=dict(C5DS['constraint'], expr=['z + a']))
        CpuPathDaeNmpc(bad, oracle_coll_gen(bad))
