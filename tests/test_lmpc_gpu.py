"""GPU parity: hilo_qp_solve (through the reference-style LMPC class and the C ABI) vs the oracle (oracle/lmpc.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.lmpc import LmpcProblem, lmpc_optimize             # noqa: E402
from tests.problems import C1, LMPC_A as A, LMPC_B as B, LMPC_DT as DT, product_lmpc   # noqa: E402


QD = np.array([[2., .5], [.5, 1.]])                            # a state weight with off-diagonal entries: H is not diagonal


@pytest.mark.parametrize('variant', ['reference', 'corrected'])
def test_c1_vs_oracle(variant):
    """BASELINE configs[0]: LMPC on the discrete double integrator (nx=2, nu=1, N=10), single instance and a batch."""
    pb = LmpcProblem(**C1, kron_bug=(variant == 'reference'))
    mpc = product_lmpc(variant)
    assert mpc._x_ind == pb.x_ind and mpc._u_ind == pb.u_ind and mpc._n_v == 32          # mpc.py:2221-2231
    np.testing.assert_array_equal(mpc._Ad.cpu().numpy(), pb.Aeq)
    np.testing.assert_array_equal(mpc._H.cpu().numpy(), pb.H)
    u = mpc.optimize([1., 1.])                                                          # tests/test_LMPC.py:10
    ref = lmpc_optimize(pb, [[1., 1.]])
    assert u.shape == (1, 1) and mpc.solver_status_code[0] == ref['status'][0] == 1
    np.testing.assert_allclose(u, ref['u'].T, rtol=1e-7, atol=1e-9)
    rng = np.random.default_rng(3)
    x0 = np.vstack([[1., 1.], rng.uniform(-2, 2, (63, 2))])
    ub = mpc.optimize(x0)
    ref = lmpc_optimize(pb, x0)
    st = mpc.solver_status_code
    assert np.array_equal(st == 1, ref['status'] == 1)          # same instances feasible / solved
    ok = st == 1
    assert ok.sum() >= 32
    v = mpc._nlp_solution['x'].cpu().numpy()
    # the oracle's active-set polish gives the vertex solution to ~1e-13; the interior-point result is within its tol
    np.testing.assert_allclose(v[ok], ref['v'][ok], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(ub[ok], ref['u'][ok], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(mpc._nlp_solution['f'].cpu().numpy()[ok], ref['f'][ok], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(mpc._nlp_solution['lam_a'].cpu().numpy()[ok], ref['lam_a'][ok], rtol=1e-5, atol=1e-6)
    # KKT in CasADi's convention at the returned point
    lam_x = mpc._nlp_solution['lam_x'].cpu().numpy()[ok]
    lam_a = mpc._nlp_solution['lam_a'].cpu().numpy()[ok]
    assert np.abs(v[ok] @ pb.H + lam_a @ pb.Aeq + lam_x).max() < 1e-7
    assert np.abs(v[ok] @ pb.Aeq.T).max() < 1e-9


def test_closed_loop_200_steps_corrected():
    """tests/test_LMPC.py:30-33: 200 closed-loop steps from x0 = [1, 1]; with the corrected input block the double
    integrator is driven to the origin."""
    import torch
    mpc = product_lmpc('corrected')
    x = torch.tensor([[1., 1.]], dtype=torch.float64, device='cuda')
    At, Bt = torch.as_tensor(A, device='cuda'), torch.as_tensor(B, device='cuda')
    for _ in range(200):
        u = mpc.optimize(x)
        x = x @ At.T + u @ Bt.T
    assert mpc.solver_status_code[0] == 1
    assert float(x.abs().max()) < 1e-6
    assert mpc._n_iterations == 200 and abs(mpc._time - 200 * DT) < 1e-9


def test_batch_1024_properties_and_errors():
    import torch
    mpc = product_lmpc('corrected')
    rng = np.random.default_rng(5)
    x0 = torch.as_tensor(rng.uniform(-1.5, 1.5, (1024, 2)), device='cuda')
    u = mpc.optimize(x0)
    assert np.all(mpc.solver_status_code == 1)
    v = mpc._nlp_solution['x']
    assert torch.equal(v[:, :2], x0) and float(v[:, 22:].abs().max()) <= 1 + 1e-9
    assert float((v @ mpc._Ad.T).abs().max()) < 1e-9
    with pytest.raises(ValueError, match="We have an issue mate, the x0 you supplied has dimension 3"):
        mpc.optimize([1., 2., 3.])
    assert mpc.optimize(torch.empty(0, 2, dtype=torch.float64, device='cuda')).shape == (0, 1)


@pytest.mark.parametrize('N', [30, 60])
def test_long_horizon_qp_in_global_workspace(N):
    """Beyond N = 22 the dense working set of the QP (n = 3N + 2, m = 2N) exceeds the LDS: it then lives in a per-instance
    global-memory workspace; parity with the oracle's QP solver on the corrected assembly."""
    rng = np.random.default_rng(3)
    x0 = rng.uniform(-1.5, 1.5, (8, 2))
    mpc = product_lmpc('corrected', N=N)
    u = mpc.optimize(x0)
    ref = lmpc_optimize(LmpcProblem(**dict(C1, N=N), kron_bug=False), x0)
    st = mpc.solver_status_code
    assert np.array_equal(st == 1, ref['status'] == 1) and (st == 1).sum() >= 6
    ok = st == 1
    np.testing.assert_allclose(u[ok], ref['u'][ok], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(mpc._nlp_solution['x'].cpu().numpy()[ok], ref['v'][ok], atol=1e-6)


@pytest.mark.parametrize('N,variant,Q', [(10, 'reference', None), (20, 'corrected', None), (20, 'reference', None),
                                         (14, 'corrected', None), (10, 'corrected', QD), (20, 'corrected', QD)])
def test_register_resident_kernel_equals_the_lds_column_kernel(N, variant, Q, monkeypatch):
    """Round 3: QPs with n, m <= 64 run on qp_solve_reg_kernel<NP, MP> (dimensions padded to 32 / 64 and 24 .. 64, factorisations in registers, inverse factors); the first
    kernel (columns through LDS, HILO_QP_LDS_COLUMNS=1) solves the same iteration - same statuses and iteration counts, solutions
    to round-off; and the oracle agrees.  N = 20 (n = 62, m = 40) exercises the two-pass column scheme of the 64-wide variant.
    Diagonal weights (Q = I: H + Sigma diagonal, the factorisation is one scaling per row) and a weight with off-diagonal entries
    (the general path: H + Sigma factored in registers)."""
    rng = np.random.default_rng(11)
    x0 = rng.uniform(-1.5, 1.5, (48, 2))
    monkeypatch.setenv('HILO_QP_DENSE', '1')                    # (the corrected block has the stage shape: keep both on the dense kernels)
    fast = product_lmpc(variant, N=N, Q=Q)
    uf = fast.optimize(x0)
    monkeypatch.setenv('HILO_QP_LDS_COLUMNS', '1')
    slow = product_lmpc(variant, N=N, Q=Q)
    us = slow.optimize(x0)
    assert not fast._qp_stages and not slow._qp_stages
    assert np.array_equal(fast.solver_status_code, slow.solver_status_code)
    ok = fast.solver_status_code == 1
    assert ok.sum() >= 24
    it_f, it_s = fast._nlp_solution['iter_count'].cpu().numpy(), slow._nlp_solution['iter_count'].cpu().numpy()
    assert np.max(np.abs(it_f[ok] - it_s[ok])) <= 1
    np.testing.assert_allclose(uf[ok], us[ok], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(fast._nlp_solution['x'].cpu().numpy()[ok], slow._nlp_solution['x'].cpu().numpy()[ok], atol=1e-8)
    np.testing.assert_allclose(fast._nlp_solution['lam_a'].cpu().numpy()[ok], slow._nlp_solution['lam_a'].cpu().numpy()[ok],
                               rtol=1e-6, atol=1e-7)
    ref = lmpc_optimize(LmpcProblem(**dict(C1, N=N, **({} if Q is None else {'Q': Q})), kron_bug=(variant == 'reference')), x0)
    both = ok & (ref['status'] == 1)
    assert both.sum() >= 24
    np.testing.assert_allclose(uf[both], ref['u'][both], rtol=1e-6, atol=1e-5)   # (longer horizons: degenerate vertices, polish)


@pytest.mark.parametrize('N,Q', [(10, None), (10, QD), (15, QD), (20, None), (40, QD)])
def test_stage_kernel_equals_the_dense_kernels(N, Q, monkeypatch):
    """Round 4: a QP with the stage shape (corrected input block) takes its Newton steps by a Riccati recursion over the stages
    (csrc/hilo_qp_ocp.h: a stage per lane, 16 lanes per instance up to N = 15, a wave beyond) - the same predictor-corrector
    iteration as the dense kernels: same statuses (incl. the infeasible starts), iteration counts within one, solution, objective
    and both multiplier sets to round-off; and the oracle agrees."""
    rng = np.random.default_rng(12)
    x0 = np.vstack([rng.uniform(-1.5, 1.5, (45, 2)), rng.uniform(-4, 4, (10, 2))])
    st = product_lmpc('corrected', N=N, Q=Q)
    assert st._qp_stages
    u1 = st.optimize(x0)
    monkeypatch.setenv('HILO_QP_DENSE', '1')
    de = product_lmpc('corrected', N=N, Q=Q)
    assert not de._qp_stages
    u2 = de.optimize(x0)
    assert np.array_equal(st.solver_status_code, de.solver_status_code)
    ok = st.solver_status_code == 1
    assert ok.sum() >= 30 and (st.solver_status_code == 3).sum() >= 1
    a, b = st._nlp_solution, de._nlp_solution
    assert np.max(np.abs(a['iter_count'].cpu().numpy() - b['iter_count'].cpu().numpy())) <= 1
    np.testing.assert_allclose(u1[ok], u2[ok], rtol=1e-8, atol=1e-9)
    for key, tol in (('x', 1e-8), ('f', 1e-9), ('lam_a', 1e-7), ('lam_x', 1e-7)):
        np.testing.assert_allclose(a[key].cpu().numpy()[ok], b[key].cpu().numpy()[ok], rtol=tol, atol=tol)
    ref = lmpc_optimize(LmpcProblem(**dict(C1, N=N, **({} if Q is None else {'Q': Q})), kron_bug=False), x0)
    both = ok & (ref['status'] == 1)
    assert both.sum() >= 30
    np.testing.assert_allclose(u1[both], ref['u'][both], rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize('mode', ['stages', 'dense', 'lds_columns'])
def test_pinned_entry_point_equals_per_instance_bound_rows(mode, monkeypatch):
    """`hilo_qp_solve_pinned` (the measured states next to ONE pair of bound rows shared by the batch - what `LMPC.optimize` calls)
    against `hilo_qp_solve` with the reference's own procedure, x_0 written into per-instance rows lbx = ubx (mpc.py:2361-2362):
    the same bytes in every result vector, on the stage kernel, the register-resident dense kernel and the LDS-column kernel."""
    import torch
    from hilo_mpc_amd import _lib
    from hilo_mpc_amd._device import ptr, stream_ptr
    if mode != 'stages':
        monkeypatch.setenv('HILO_QP_DENSE', '1')
    if mode == 'lds_columns':
        monkeypatch.setenv('HILO_QP_LDS_COLUMNS', '1')
    mpc = product_lmpc('corrected')
    assert bool(mpc._qp_stages) == (mode == 'stages')
    rng = np.random.default_rng(5)
    x0 = np.vstack([rng.uniform(-1.5, 1.5, (60, 2)), rng.uniform(-4, 4, (6, 2))])
    mpc.optimize(x0)
    a = {k: v.clone() for k, v in mpc._nlp_solution.items()}
    B, n, m, dev = x0.shape[0], mpc._n_v, mpc._n_g, mpc._dev
    x = torch.as_tensor(x0, device=dev)
    lb, ub = mpc._v_lb.expand(B, -1).contiguous(), mpc._v_ub.expand(B, -1).contiguous()
    lb[:, :2] = x
    ub[:, :2] = x
    out = dict(x=torch.empty(B, n, dtype=torch.float64, device=dev), f=torch.empty(B, dtype=torch.float64, device=dev),
               lam_a=torch.empty(B, m, dtype=torch.float64, device=dev), lam_x=torch.empty(B, n, dtype=torch.float64, device=dev),
               status=torch.empty(B, dtype=torch.int32, device=dev), iter_count=torch.empty(B, dtype=torch.int32, device=dev))
    _lib.check(_lib.lib().hilo_qp_solve(mpc._handle, B, ptr(mpc._H), 0, ptr(mpc._g), 0, ptr(mpc._Ad), 0, ptr(lb), ptr(ub), n,
                                        ptr(mpc._beq), ptr(mpc._beq), 0, ptr(out['x']), ptr(out['f']), ptr(out['lam_a']),
                                        ptr(out['lam_x']), ptr(out['status']), ptr(out['iter_count']), stream_ptr(dev)))
    assert (a['status'] == 1).sum() >= 55 and (a['status'] == 3).sum() >= 1
    for k in out:
        assert torch.equal(a[k], out[k]), k
    # argument checks of the new entry: shared rows only together with pinned values
    with pytest.raises(Exception, match='per instance'):
        _lib.check(_lib.lib().hilo_qp_solve(mpc._handle, B, ptr(mpc._H), 0, ptr(mpc._g), 0, ptr(mpc._Ad), 0, ptr(mpc._v_lb), ptr(mpc._v_ub), 0,
                                            ptr(mpc._beq), ptr(mpc._beq), 0, ptr(out['x']), ptr(out['f']), ptr(out['lam_a']),
                                            ptr(out['lam_x']), ptr(out['status']), ptr(out['iter_count']), stream_ptr(dev)))


def test_stage_kernel_on_a_four_state_two_input_system(monkeypatch):
    """nx = 4, nu = 2 (two coupled double integrators), N = 12, bounds on states and inputs, B = 64: stage kernel against the
    dense register kernel; single instance too (the launch of BASELINE configuration 1's shape: one QP)."""
    from hilo_mpc_amd import LMPC, Model
    dt = .2
    A4 = np.array([[1., dt, 0., 0.], [0., 1., 0., 0.], [.05, 0., 1., dt], [0., 0., -.02, 1.]])
    B4 = np.array([[.5 * dt * dt, 0.], [dt, 0.], [0., .5 * dt * dt], [.1 * dt, dt]])

    def make():
        mpc = LMPC(Model('lti', A=A4, B=B4).setup(dt=dt))
        mpc.Q, mpc.R, mpc.P = np.diag([1., .1, 2., .1]), np.array([[1., .2], [.2, .5]]), 3 * np.eye(4)
        mpc.horizon = 12
        mpc.set_box_constraints(x_lb=[-3, -2, -3, -2], x_ub=[3, 2, 3, 2], u_lb=[-1, -1.5], u_ub=[1, 1.5])
        mpc.setup(kron_variant='corrected')
        return mpc
    rng = np.random.default_rng(5)
    x0 = rng.uniform(-1, 1, (64, 4))
    st = make()
    assert st._qp_stages
    u1 = st.optimize(x0)
    st1 = st.solver_status_code.copy()
    one = st.optimize(x0[3])
    monkeypatch.setenv('HILO_QP_DENSE', '1')
    de = make()
    u2 = de.optimize(x0)
    assert np.array_equal(st1, de.solver_status_code) and np.all(de.solver_status_code == 1)
    np.testing.assert_allclose(u1, u2, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(np.ravel(one), u2[3], rtol=1e-8, atol=1e-9)
    assert (np.abs(u1[:, 0]) > 1 - 1e-6).sum() >= 3                       # input bounds active somewhere


@pytest.mark.parametrize('lds_columns', [False, True])
def test_infeasible_states_are_reported_early(lds_columns, monkeypatch):
    """Measured states from which the box cannot be kept: status 3 (`Infeasible_Problem_Detected`) after a few iterations by
    OOQP's termination rule, in both kernels, on exactly the instances the oracle reports."""
    if lds_columns:
        monkeypatch.setenv('HILO_QP_LDS_COLUMNS', '1')
    rng = np.random.default_rng(8)
    x0 = rng.uniform(-4, 4, (256, 2))
    mpc = product_lmpc('corrected')
    mpc.optimize(x0)
    ref = lmpc_optimize(LmpcProblem(**C1, kron_bug=False), x0)
    st = mpc.solver_status_code
    assert np.array_equal(st, ref['status'])
    assert (st == 3).sum() >= 20 and (st == 1).sum() >= 100 and set(np.unique(st)) == {1, 3}
    it = mpc._nlp_solution['iter_count'].cpu().numpy()
    assert it[st == 3].max() <= 15 and np.max(np.abs(it - ref['iters'])) <= 1


def test_models_written_as_expressions():
    """A linear model written as expressions gives the QP of the same system handed over as matrices (tests/test_LMPC.py:8-19),
    bit for bit; the linearised bicycle of tests/test_LMPC.py:58-116 (discretised, linearised about the origin) and its variant
    with constant parameters (:118-168) set up and solve."""
    from hilo_mpc_amd import LMPC, Model
    from tests.test_linearize import _bicycle, _double_integrator
    ref = product_lmpc('corrected')
    mpc = LMPC(_double_integrator(DT))
    mpc.Q, mpc.R, mpc.horizon = np.eye(2), 1, 10
    mpc.set_box_constraints(x_lb=[-5, -5], x_ub=[5, 5], u_lb=[-1], u_ub=[1])
    mpc.setup(kron_variant='corrected')
    np.testing.assert_array_equal(mpc._Ad.cpu().numpy(), ref._Ad.cpu().numpy())
    x0 = np.random.default_rng(5).uniform(-1.5, 1.5, (16, 2))
    np.testing.assert_array_equal(mpc.optimize(x0), ref.optimize(x0))
    for with_parameters in (False, True):
        ml = _bicycle(with_parameters).linearize()
        ml.setup(dt=.05)
        if with_parameters:
            ml.set_initial_parameter_values(p=[1.4, 1.8])
        ml.set_equilibrium_point(x_eq=[0, 0, 0, 0], u_eq=[0, 0])
        mpc = LMPC(ml)
        mpc.horizon = 10
        mpc.Q, mpc.R = np.eye(4), np.eye(2)
        mpc.setup(kron_variant='corrected')
        cp = [1.4, 1.8] if with_parameters else None
        u = mpc.optimize([.5, 0, 0, 0], cp=cp)                      # tests/test_LMPC.py:113, :167
        assert u.shape == (2, 1) and mpc.solver_status_code[0] == 1
        X, U = mpc.return_prediction()
        A, B, _ = ml.system_matrices()
        np.testing.assert_allclose(X[0][:, 1:], A @ X[0][:, :-1] + B @ U[0], atol=1e-9)     # the prediction obeys x+ = A x + B u
        if with_parameters:
            with pytest.raises(ValueError, match="constant parameter"):
                mpc.optimize([.5, 0, 0, 0])
            # constant parameters alone and the DEFAULT setup: the reference stays on its `kron(B, I_N)` branch (mpc.py:2241-2243),
            # which is not the stage shape for nx > 1 - the stage kernel must not be declared (it reads only the block-diagonal
            # positions of the input block), and the QP is the one of the same matrices handed over directly
            df = LMPC(ml)
            df.horizon = 10
            df.Q, df.R = np.eye(4), np.eye(2)
            df.setup()
            ud = df.optimize([.5, 0, 0, 0], cp=cp)
            assert not df._qp_stages and df.solver_status_code[0] == 1
            mat = LMPC(Model('lti', A=A, B=B).setup(dt=.05))
            mat.horizon = 10
            mat.Q, mat.R = np.eye(4), np.eye(2)
            mat.setup()
            assert not mat._qp_stages
            np.testing.assert_array_equal(df._Ad.cpu().numpy(), mat._Ad.cpu().numpy())
            np.testing.assert_array_equal(ud, mat.optimize([.5, 0, 0, 0]))
            assert not np.allclose(ud, u, atol=1e-6)     # ... and a different problem than the block-diagonal one
            # tests/test_LMPC.py:175-188: one length varies along the horizon; held constant it is the constant-parameter problem
            # in the block-diagonal ("corrected") form of the input block
            tv = LMPC(ml)
            tv.horizon = 10
            tv.Q, tv.R = np.eye(4), np.eye(2)
            tv.set_time_varying_parameters(names=['lf'])
            tv.setup()
            ut = tv.optimize([.5, 0, 0, 0], cp=[1.4], tvp={'lf': [1.8] * 10})
            np.testing.assert_array_equal(ut, u)
            ut = tv.optimize([1, 1, 0, 0], cp=[1.4], tvp={'lf': [1.8] * 5 + [1.4] * 5})
            assert tv.solver_status_code[0] == 1
            X, U = tv.return_prediction()
            for k in range(10):
                Ak, Bk, _ = ml.system_matrices(p=[1.4, 1.8 if k < 5 else 1.4])
                np.testing.assert_allclose(X[0][:, k + 1], Ak @ X[0][:, k] + Bk @ U[0][:, k], atol=1e-9)


def test_nmpc_and_lmpc_agree_on_a_linear_model_with_a_time_varying_parameter():
    """tests/test_LMPC.py:189-240 (`test_compare_NMPC_LMPC`): x+ = [[-1, 2 p], [0, -1]] x + u with p changing along the stored series,
    the same quadratic cost and input bound in both controllers - two transcriptions and two solvers (interior point on the
    run-time compiled model, the QP kernel on the per-stage matrices) of one problem give the same input, step after step."""
    from hilo_mpc_amd import LMPC, NMPC, Model
    m = Model(discrete=True)
    x = m.set_dynamical_states(['x_1', 'x_2'])
    u = m.set_inputs(['u_1', 'u_2'])
    q = m.set_parameters(['p'])
    m.set_dynamical_equations([-1. * x[0] + 2. * q[0] * x[1] + u[0], -1. * x[1] + u[1]])
    m.setup(dt=1.)
    tvp = 12 * [1.] + 30 * [0.]
    nmpc = NMPC(m)
    nmpc.horizon = 10
    nmpc.quad_stage_cost.add_states(names=['x_1', 'x_2'], weights=[1, 1])
    nmpc.quad_terminal_cost.add_states(names=['x_1', 'x_2'], weights=[1, 1])
    nmpc.quad_stage_cost.add_inputs(names=['u_1', 'u_2'], weights=[1, 1])
    nmpc.set_box_constraints(u_ub=[0.5, 10])
    nmpc.set_time_varying_parameters(names=['p'], values={'p': tvp})
    nmpc.set_solver_opts({'ipopt.tol': 1e-10})
    nmpc.setup()
    lmpc = LMPC(m)
    lmpc.horizon = 10
    lmpc.Q, lmpc.P, lmpc.R = np.eye(2), np.eye(2), np.eye(2)
    lmpc.set_time_varying_parameters(names=['p'], values={'p': tvp})
    lmpc.set_box_constraints(u_ub=[0.5, 10])
    lmpc.setup()
    xi = np.array([[1., 2.], [-.5, 1.5]])
    active = False
    for i in range(16):
        un, ul = nmpc.optimize(xi), lmpc.optimize(xi)
        assert np.all(nmpc.solver_status_code == 1) and np.all(lmpc.solver_status_code == 1)
        np.testing.assert_allclose(un, ul, rtol=1e-6, atol=1e-6)
        active |= bool(np.any(ul[:, 0] > .5 - 1e-6))
        Ai, Bi, _ = m.system_matrices(p=[tvp[i]])
        xi = xi @ Ai.T + ul @ Bi.T
    assert active                                              # the input bound was active on the way
