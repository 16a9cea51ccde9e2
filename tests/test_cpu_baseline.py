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
    with pytest.raises(RuntimeError, match="chemostat4 and pendulum4"):
        CpuNmpc(NmpcProblem(models.get('cstr3'), dt=1., N=4))
