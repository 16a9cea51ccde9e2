"""CPU: host-side contract of the reference-style front end (no device needed before setup()): the compact setters fill
the same objects as the long form (mpc.py:1064-1131, optimizer.py:1154-1178), argument validation raises what the reference
raises, options follow the reference's allow-lists (optimizer.py:1388-1474)."""
import numpy as np
import pytest

from hilo_mpc_amd import NMPC, Model, expr


def _model():
    return Model('chemostat4').discretize('erk', order=4).setup(dt=0.5)


def test_compact_cost_setters_equal_the_long_form():
    m = _model()
    a, b = NMPC(m), NMPC(m)
    a.set_quadratic_stage_cost(states=['X', 'S'], cost_states=[1., 2.], states_references=[5., 10.], inputs=['DS'],
                               cost_inputs=[.1], inputs_references=[.2])
    a.set_quadratic_terminal_cost(states=['X'], cost=[3.], references=[5.])
    b.quad_stage_cost.add_states(names=['X', 'S'], weights=[1., 2.], ref=[5., 10.])
    b.quad_stage_cost.add_inputs(names=['DS'], weights=[.1], ref=[.2])
    b.quad_terminal_cost.add_states(names=['X'], weights=[3.], ref=[5.])
    for ca, cb in ((a.quad_stage_cost, b.quad_stage_cost), (a.quad_terminal_cost, b.quad_terminal_cost)):
        assert len(ca._terms) == len(cb._terms)
        for (k1, i1, W1, r1), (k2, i2, W2, r2) in zip(ca._terms, cb._terms):
            assert (k1, i1, r1) == (k2, i2, r2) and np.array_equal(W1, W2)
    c = NMPC(m)
    c.set_quadratic_stage_cost(states=['X'], cost_states=[1.])            # inputs left out: no input term, no error
    assert [t[0] for t in c.quad_stage_cost._terms] == ['states']


def test_compact_constraint_setters():
    m = _model()
    n = NMPC(m)
    n.set_stage_constraints(m.x['X'] * m.x['S'], lb=[0.], ub=[60.], is_soft=True, weight=[[100.]])
    sc = n.stage_constraint
    assert sc.is_set and sc.size == 1 and sc.is_soft and sc.lb == [0.] and sc.ub == [60.] and sc.max_violation is None
    assert np.array_equal(np.asarray(sc.weight), [[100.]])
    n.set_terminal_constraints([m.x['X'] + m.x['P'], m.x['S']], lb=[-np.inf, 30.], ub=[1., np.inf])
    tc = n.terminal_constraint
    assert tc.is_set and tc.size == 2 and not tc.is_soft and tc.ub == [1., np.inf]
    n.set_stage_constraints(None)
    assert not n.stage_constraint.is_set
    with pytest.raises(TypeError):
        n.set_stage_constraints('X*S')                                      # strings are not expressions
    with pytest.raises(TypeError):
        n.stage_constraint.is_soft = 1


def test_cost_argument_errors():
    m = _model()
    n = NMPC(m)
    with pytest.raises(ValueError, match="does not exist"):
        n.quad_stage_cost.add_states(names=['nope'], weights=[1.])
    with pytest.raises(ValueError, match="weights"):
        n.quad_stage_cost.add_states(names=['X'], weights=None)
    with pytest.raises(ValueError, match="dimensions"):
        n.quad_stage_cost.add_states(names=['X', 'S'], weights=[1., 1.], ref=[1.])
    with pytest.raises(TypeError, match="same number of bounds"):       # optimizer.py:1318-1386 raises TypeError
        n.set_box_constraints(x_ub=[1., 2.])
    with pytest.raises(ValueError):
        n.set_scaling(u_scaling=[1.])


def test_expression_symbols_and_depth():
    m = _model()
    with pytest.raises(KeyError):
        m.x['nope']
    e = m.x['X']
    for _ in range(9):
        e = 1. + (e * e)                                                   # right-nested: needs a deeper stack
    deep = m.x['S']
    for _ in range(10):
        deep = m.x['X'] + (m.x['S'] * deep)
    with pytest.raises(ValueError, match="too deep"):
        deep.program()
    assert (m.x['X'] ** 2).program()[0] == 4.                               # [len | VARX 0 | SQ 0]
    assert repr(m.x['X'] ** 2.5) == 'exp(mul(2.5, log(X)))'                  # general power of a positive base: exp(b log a)
    assert len((m.x['X'] ** 2.5).program()) == 1 + 2 * 5
    with pytest.raises(NotImplementedError):
        m.x['X'] ** 'two'
    with pytest.raises(ValueError, match="path variable"):
        expr.Expr('theta', name='theta').program()


def test_mhe_host_contract():
    """MHE front end before setup(): same failure modes as mhe.py (horizon / cost checks), unsupported branches refuse."""
    from hilo_mpc_amd import MHE
    m = Model('chemostat4').discretize('erk', order=4).setup(dt=0.5)
    mhe = MHE(m)
    with pytest.raises(ValueError, match="horizon"):
        mhe.horizon = 0
    with pytest.raises(ValueError, match="horizon length"):
        mhe.setup()
    mhe.horizon = 5
    with pytest.raises(ValueError, match="cost function"):
        mhe.setup()
    with pytest.raises(ValueError, match="same dimension"):
        mhe.quad_arrival_cost.add_states(weights=[1., 1., 1., 1.], guess=[1., 2.])
    with pytest.raises(NotImplementedError):
        mhe.quad_stage_cost.add_inputs(names=['DS'], weights=[1.])
    with pytest.raises(TypeError, match="same number of bounds"):
        mhe.set_box_constraints(x_lb=[0., 0.])
    with pytest.raises(RuntimeError, match="setup"):
        mhe.add_measurements([1., 2.])
    with pytest.raises(ValueError, match="setup"):
        mhe.estimate()
    # mhe.py:792-860, :911-932, :1070-1087, :1214-1246
    with pytest.raises(ValueError, match="The option named nope does not exist"):
        mhe.set_nlp_options({'nope': 1})
    with pytest.raises(ValueError, match="only allowed values"):
        mhe.set_nlp_options(arrival_guess_update='sometimes')
    mhe.set_nlp_options({'arrival_guess_update': 'smoothing', 'print_level': 0})
    with pytest.raises(ValueError, match="is not in the model0 parameter"):
        mhe.set_time_varying_parameters(['nope'])
    with pytest.warns(UserWarning, match="do not enter the estimation problem"):      # mhe.py:677: the tv_p slot is never read
        mhe.set_time_varying_parameters(['Sf'])
    mhe.set_time_varying_parameters()
    with pytest.raises(ValueError, match="you must pass"):
        mhe.set_aux_nonlinear_constraints(aux_nl_const=m.x['X'], ub=[1.])
    mhe.set_aux_nonlinear_constraints()
    with pytest.raises(TypeError, match="has_state_noise accepts True or False"):
        mhe.has_state_noise = 1
    assert mhe.has_state_noise is False
    mhe.quad_stage_cost.add_state_noise(weights=[1., 1., 1., 1.])
    assert mhe.has_state_noise is True
    with pytest.warns(UserWarning, match="no mpc solution"):
        assert mhe.return_mhe_estimation() == (None, None)
    # layout of the window in the decision vector (mhe.py:614-655): [p | x_0..x_N | w_0..w_{N-1}], un-scaled on return
    import torch
    mhe._x_ind = [list(range(k * 4, (k + 1) * 4)) for k in range(6)]
    mhe._w_ind = [list(range(24 + k * 4, 24 + (k + 1) * 4)) for k in range(5)]
    mhe._sx, mhe._w_scaling = np.array([1., 10., 1., 1.]), [2., 1., 1., 1.]
    v = np.arange(2 * 44, dtype=float).reshape(2, 44)
    mhe._nlp_solution = {'x': torch.as_tensor(v)}
    X, W = mhe.return_mhe_estimation()
    assert X.shape == (2, 4, 6) and W.shape == (2, 4, 5)
    assert X[1, 1, 2] == v[1, 2 * 4 + 1] * 10. and W[0, 0, 3] == v[0, 24 + 3 * 4] * 2.


def test_lmpc_host_contract():
    from hilo_mpc_amd import LMPC
    with pytest.raises(TypeError, match="nonlinear"):
        LMPC(Model('chemostat4').discretize('erk', order=4).setup(dt=0.5))
    m = Model('lti', A=[[1., 1.], [0., 1.]], B=[[0.5], [1.]]).setup(dt=1.)
    lmpc = LMPC(m)
    for f in ('set_stage_constraints', 'set_custom_constraints_function', 'set_initial_guess'):      # mpc.py:2396-2406
        with pytest.raises(NotImplementedError, match=f"The method {f} is not available for LMPC."):
            getattr(lmpc, f)()
    with pytest.raises(ValueError, match="2x2"):
        lmpc.Q = np.eye(3)
    lmpc.Q, lmpc.R, lmpc.P = np.eye(2), [[1.]], np.eye(2)
    with pytest.raises(ValueError, match="horizon"):
        lmpc.horizon = -1
    with pytest.raises(ValueError, match="horizon length"):
        lmpc.setup()
    lmpc.horizon = 10
    with pytest.raises(ValueError, match="does no exist"):
        lmpc.setup(solver='osqp')
    with pytest.raises(ValueError, match="kron_variant"):
        lmpc.setup(kron_variant='other')
    with pytest.raises(TypeError, match="same number of bounds"):
        lmpc.set_box_constraints(u_ub=[1., 2.])
    with pytest.raises(ValueError, match="setup"):
        lmpc.optimize([0., 0.])


def test_nmpc_requires_setup_and_options_follow_the_allow_lists():
    m = _model()
    n = NMPC(m)
    with pytest.raises(ValueError, match="setup"):
        n.optimize([1., 1., 1., 1.])
    n.quad_stage_cost.add_states(names=['X'], weights=[1.])
    n.horizon = 5
    with pytest.raises(ValueError):
        n.set_nlp_options({'integration_method': 'euler_backward'})      # not in the allow-list (optimizer.py:1388-1474)
    with pytest.raises((ValueError, KeyError, TypeError)):
        n.set_nlp_options({'no_such_option': 1})


# ---- round 2: host logic of the new front-end pieces (no device needed) --------------------------------------------------------
def test_measurement_cost_expression_matches_its_definition():
    """`add_measurements` (modeling.py:385-408): the compiled expression is (h(x, u) - ref / s_y)^T W (.) on the model's
    measurement equations, evaluated on the (scaled) NLP variables like every cost of the reference."""
    from hilo_mpc_amd.symdiff import Dag
    from tests.problems import symbolic_model
    m = symbolic_model('cstr3').setup(dt=1.)
    n = NMPC(m)
    n.quad_stage_cost.add_measurements(names=['y_0'], weights=[3.], ref=[.004])
    with pytest.raises(ValueError, match="does not exist"):
        n.quad_stage_cost.add_measurements(names=['nope'], weights=[1.])
    with pytest.raises(NotImplementedError):
        n.quad_stage_cost.add_measurements(names=['y_0'], weights=[1.], trajectory_tracking=True)
    e = n.quad_stage_cost._measurement_cost([2.])
    g = Dag()
    node = g.from_expr(e)
    x, u = [.5, .5, 430.], [5e4]
    r = 5000. * np.exp(-1e4 / (1.987 * x[2])) * x[0] - 1e6 * np.exp(-1.5e4 / (1.987 * x[2])) * x[1]
    np.testing.assert_allclose(g.evaluate([node], x, u, [])[0], 3. * (r - .004 / 2.) ** 2, rtol=1e-13)
    assert NMPC(m).quad_stage_cost._measurement_cost(None) is None
    # a zoo model without expression definitions has nothing to build the term from
    z = NMPC(Model('robot6').discretize('rk4').setup(dt=.1))
    z.quad_stage_cost.add_measurements(names=z._model.measurement_names[:1], weights=[1.])
    with pytest.raises(NotImplementedError, match="measurement equations as expressions"):
        z.quad_stage_cost._measurement_cost(None)


def test_algebraic_state_declarations():
    from tests.problems import symbolic_model
    m = symbolic_model('pendulum4_dae')
    assert m.n_z == 1 and m.algebraic_state_names == ['y'] and len(m._alg) == 1
    with pytest.raises(ValueError, match="1 algebraic states but 2"):
        m.set_algebraic_equations([m.z[0], m.z[0]])
    zoo = Model('chemostat4')
    with pytest.raises(RuntimeError, match="device zoo"):
        zoo.set_algebraic_states(['z'])
    n = NMPC(m.setup(dt=.1))
    n.set_initial_guess(x_guess=[0, 0, 0, 0], u_guess=0., z_guess=1.4)
    assert n._z_guess == [1.4]
    n.set_box_constraints(z_lb=[-1.], z_ub=[2.])                       # finite bounds: rows on z at the collocation points (tests/test_dae_gpu.py)
    assert n._z_lb == [-1.] and n._z_ub == [2.]
    with pytest.raises(TypeError):
        n.set_box_constraints(z_lb=[-1., 0.])
    n.set_box_constraints(z_lb=[-np.inf], z_ub=[np.inf])               # the reference's defaults


def test_simple_control_loop_sequence_with_a_stub_controller():
    """control_loop.py:343-397: optimize on the plant state, plant step, observer step - in that order, `steps` times."""
    from hilo_mpc_amd import SimpleControlLoop
    calls = []

    class Ctl:
        solver_status_code = np.array([1, 1])

        def optimize(self, x, cp=None):
            calls.append(('opt', x.copy(), cp))
            return -0.5 * x[:, :1]

    class Obs:
        def estimate(self, y=None, u=None):
            calls.append(('est', y.copy(), u.copy()))
            return y

    loop = SimpleControlLoop(lambda x, u, p: x + u, Ctl(), Obs())
    sol = loop.run(3, np.array([[2., 4.], [1., -1.]]), p=[7.], measure=lambda x: x[:, :1])
    assert [c[0] for c in calls] == ['opt', 'est'] * 3 and calls[0][2] == [7.]
    assert sol['x'].shape == (4, 2, 2) and sol['u'].shape == (3, 2, 1) and len(sol['estimates']) == 3
    np.testing.assert_allclose(sol['x'][1], [[1., 3.], [.5, -1.5]])
    with pytest.raises(TypeError):
        SimpleControlLoop(lambda x, u, p: x, object())


def test_backend_selection_is_validated(monkeypatch):
    import os
    assert os.environ.get('HILO_NMPC_BACKEND', 'auto') in ('auto', 'precompiled', 'runtime')


def test_trajectory_reference_as_function_of_time_equals_the_sampled_trajectory(monkeypatch):
    """`add_states(..., ref=f(t), trajectory_tracking=True)` with `t = nmpc.get_time_variable()` (mpc.py:232-246, :1055-1062): the
    per-stage reference table of a solve holds f(t_0 + k dt) - the same numbers as the sampled trajectory passed per call; the
    controller's clock advances by the sampling interval per optimize (mpc.py:850)."""
    import numpy as np
    from hilo_mpc_amd import NMPC, Model
    from hilo_mpc_amd.expr import sin
    monkeypatch.setenv('HILO_JIT_COMPILE_ONLY', '1')         # setup() without a GPU (nothing is compiled for this zoo problem)
    N, dt = 6, .5

    def build(fun):
        m = Model('chemostat4').discretize('rk4').setup(dt=dt)
        nmpc = NMPC(m)
        t = nmpc.get_time_variable()
        kw = dict(ref=[1. + .05 * t + .1 * sin(2. * t)]) if fun else {}
        nmpc.quad_stage_cost.add_states(names=['P'], weights=[10.], trajectory_tracking=True, **kw)
        nmpc.quad_stage_cost.add_inputs(names=['DS', 'DI'], weights=[.1, .1])
        nmpc.quad_terminal_cost.add_states(names=['P'], weights=[10.], trajectory_tracking=True, **kw)
        nmpc.horizon = N
        nmpc.set_scaling(x_scaling=[1., 10., 2., 1.])
        nmpc.setup(options={'integration_method': 'discrete'})
        return nmpc

    f, s = build(True), build(False)
    assert f.quad_stage_cost.name_open_varying_trajectories == [] and s.quad_stage_cost.name_open_varying_trajectories == ['P']
    traj = [1. + .05 * (k * dt) + .1 * np.sin(2. * k * dt) for k in range(40)]
    cp = [100., 4., 1., 0.]
    for it in range(3):
        for c in (f, s):
            c._time, c._n_iterations = it * dt, it              # what optimize() does after each solve
        tf = f._stage_table(cp, None, {})
        ts = s._stage_table(cp, None, {'ref_sc': {'P': traj}, 'ref_tc': {'P': traj}})
        np.testing.assert_allclose(tf, ts, rtol=1e-15, atol=0)
        assert tf[0, 2] == traj[it] / 2. and tf[N, 2] == traj[it + N] / 2.
    with pytest.raises(ValueError, match="already been provided as a function|I cannot find the variable"):
        f._stage_table(cp, None, {'ref_sc': {'P': traj}})
    with pytest.raises(ValueError, match="only be a function of the time variable"):
        m = Model('chemostat4').discretize('rk4').setup(dt=dt)
        n2 = NMPC(m)
        n2.quad_stage_cost.add_states(names=['P'], weights=[1.], ref=[m.x['S'] * 2.], trajectory_tracking=True)
    m = Model('chemostat4').setup(dt=dt)                        # continuous model, continuous objective: refused
    n3 = NMPC(m)
    n3.quad_stage_cost.add_states(names=['P'], weights=[1.], ref=[n3.get_time_variable()], trajectory_tracking=True)
    n3.horizon = 4
    with pytest.raises(NotImplementedError, match="discrete objective"):
        n3.setup()


def test_measurement_box_constraints_become_stage_and_terminal_constraints(monkeypatch):
    """`set_box_constraints(y_ub=, y_lb=)` (mpc.py:703-708): an extra stage and terminal constraint on the measurement
    equations - for a zoo model held as expressions and for a model written as expressions; compiled without a GPU."""
    import numpy as np
    from hilo_mpc_amd import NMPC, Model
    from tests.problems import symbolic_model
    monkeypatch.setenv('HILO_JIT_COMPILE_ONLY', '1')
    for m in (Model('chemostat4').discretize('rk4').setup(dt=1.), symbolic_model('chemostat4').discretize('rk4').setup(dt=1.)):
        nmpc = NMPC(m)
        nmpc.quad_stage_cost.add_states(names=['P'], weights=[10.], ref=[1.])
        nmpc.quad_stage_cost.add_inputs(names=['DS', 'DI'], weights=[.1, .1])
        nmpc.horizon = 5
        nmpc.set_box_constraints(y_ub=[.5, 2.], u_lb=[0., 0.])
        sc, tc = nmpc.stage_constraint, nmpc.terminal_constraint
        assert sc.is_set and tc.is_set and sc.size == 2 and sc.ub == [.5, 2.] and sc.lb is None and sc._name == 'measurement_constraint'
        assert repr(sc.constraint[0]) == 'X' and repr(sc.constraint[1]) == 'P'          # yX = X, yP = P
        nmpc.setup(options={'integration_method': 'discrete'})
        assert 'NEXPR = 2' in nmpc._user_source and 'NTEXPR = 2' in nmpc._user_source
    with pytest.raises(TypeError, match="The model has 2 measurements"):
        nmpc.set_box_constraints(y_ub=[1.])
    with pytest.raises(NotImplementedError, match="measurement equations as expressions"):
        NMPC(Model('bioreactor3').discretize('rk4').setup(dt=1.)).set_box_constraints(y_lb=[0., 0.])


def test_controller_accessors_of_the_reference():
    """optimizer.py:1508-1768, mpc.py:1926-1931: clock, bounds, time variable, sampling interval, solver name; problems that do
    not fit the stage-wise solver are refused with a reason."""
    from hilo_mpc_amd import NMPC, Model
    nmpc = NMPC(Model('chemostat4').discretize('rk4').setup(dt=.5))
    assert nmpc.current_time == 0. and nmpc.initial_time == 0. and nmpc.n_iterations == 0 and nmpc.n_tvp == 0
    assert nmpc.sampling_interval == .5 and not nmpc.is_setup() and nmpc.time_var == []
    nmpc.set_box_constraints(x_lb=[0., 0., 0., 0.], u_ub=[1., 1.])
    assert nmpc.x_lb == [0., 0., 0., 0.] and nmpc.x_ub is None and nmpc.u_ub == [1., 1.] and nmpc.u_lb is None
    t = nmpc.get_time_variable()
    assert nmpc.time_var is t
    nmpc.set_sampling_interval(2)
    assert nmpc.sampling_interval == 2
    with pytest.raises(TypeError, match="Sampling interval must be a float."):
        nmpc.set_sampling_interval('1')
    nmpc.set_time_varying_parameters(['Sf'])
    assert nmpc.n_tvp == 1
    nmpc.set_nlp_solver('ipopt')
    nmpc.reset_solution()
    nmpc.minimize_final_time(weight=2)                                  # mpc.py:859-866 (solved in tests/test_coll_gpu.py)
    assert nmpc._minimize_final_time_flag and nmpc._minimize_final_time_weight == 2.
    nmpc.horizon = 5
    with pytest.raises(NotImplementedError, match="continuous model written as expressions"):
        nmpc.setup()                                                    # (this controller sits on a model of the device zoo)
    # optimizer.py:1180-1208: accepted (offloaded for stage-additive functions, hard and soft, tests/test_custom_gpu.py)
    nmpc.set_custom_constraints_function(fun=lambda v, xi, ui: v[0], ub=3)
    assert nmpc._custom_constraint_flag and nmpc._custom_constraint_size == 1 and not nmpc._custom_constraint_is_soft_flag
    assert nmpc._custom_constraint_fun_lb == [-np.inf] and nmpc._custom_constraint_fun_ub == [3.]
    nmpc.set_custom_constraints_function(fun=lambda v, xi, ui: [v[0], v[1]], lb=[0, 0], ub=[3, 4], soft=True, max_violation=2.)
    assert nmpc._custom_constraint_is_soft_flag and list(nmpc._custom_constraint_maximum_violation) == [2., 2.]
    with pytest.raises(ValueError, match="max_violation must be one value or one per"):
        nmpc.set_custom_constraints_function(fun=lambda v, xi, ui: [v[0], v[1]], lb=[0, 0], ub=[3, 4], soft=True, max_violation=[1., 2., 3.])
    with pytest.raises(TypeError, match="must be a function"):
        nmpc.set_custom_constraints_function(fun=None)


def test_model_as_plant_host_contract(monkeypatch):
    """`set_initial_conditions` / `simulate` / `solution` (dynamic_model.py:3360-3400, :3911-4000) around the device plant step
    (stubbed here: no GPU): error behaviour, orientation of the stored states, the key language of the solution."""
    import numpy as np
    from hilo_mpc_amd import Model
    m = Model('chemostat4').discretize('rk4')
    with pytest.raises(RuntimeError, match="Model is not set up. Run Model.setup\\(\\) before setting the initial conditions."):
        m.set_initial_conditions([.1, 40., 0., 0.])
    m.setup(dt=.5)
    with pytest.raises(RuntimeError, match="No initial dynamical states found"):
        m.simulate(u=[.1, .2], p=[100., 4., 1., 0.])
    with pytest.raises(ValueError, match="Dimension mismatch"):
        m.set_initial_conditions([1., 2., 3.])
    m.set_initial_conditions([.1, 40., 0., 0.])
    calls = []

    def step(x, u=None, p=None, device_index=None):
        calls.append((np.array(x), None if u is None else np.array(u)))
        return x + 1., x[:, [0, 2]]
    monkeypatch.setattr(m, 'step', step)
    m.simulate(u=[.1, .2], p=[100., 4., 1., 0.], steps=3)
    assert len(calls) == 3 and calls[0][1].shape == (1, 2)
    sol = m.solution
    np.testing.assert_array_equal(sol['x:f'], np.array([[3.1], [43.], [3.], [3.]]))          # column vector like the reference's DM
    np.testing.assert_array_equal(sol['x:0'], np.array([[.1], [40.], [0.], [0.]]))
    assert sol['x'].shape == (4, 4) and sol['y'].shape == (2, 3) and sol['t:f'] == 1.5
    c = m.copy(setup=False)
    assert c._sim is None and m._sim is not None
    # a batch of plants keeps the batch axis
    m.set_initial_conditions(np.ones((5, 4)))
    m.simulate(u=np.zeros((5, 2)), p=[100., 4., 1., 0.])
    assert m.solution['x:f'].shape == (5, 4) and m.solution['x'].shape == (2, 5, 4)


def test_model_step_passes_the_arguments_of_the_c_signature(monkeypatch):
    """`Model.step` against a stand-in for `hilo_pf_function` that READS its pointer arguments the way include/hilo_hip.h
    declares them (batch, n_samples, X, y, up, up_stride, w, v, R, r_stride, X_prop, Y, q, stream) and answers with the oracle's
    model - argument order, strides and shapes of the host call, without a GPU."""
    import ctypes as C
    import numpy as np
    import torch
    from hilo_mpc_amd import Model, _lib
    from oracle import models as omodels
    m = Model('chemostat4').discretize('rk4').setup(dt=.5)
    om = omodels.get('chemostat4').discretize(4)

    class H:
        _dev, _handle, _n_p, _n_y = torch.device('cpu'), 1234, 4, 2
    monkeypatch.setattr(m, '_plant_handle', lambda device_index=None: H)
    monkeypatch.setattr('hilo_mpc_amd._device.stream_ptr', lambda dev: 0)

    def arr(ptr, n):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(n,))

    class Lib:
        @staticmethod
        def hilo_pf_function(h, B, N, X, y, up, us, w, v, R, rs, Xp, Y, q, stream):
            assert (h, N, us, rs) == (1234, 1, 6, 0)
            x = arr(X, B * 4).reshape(B, 4)
            upv = arr(up, B * 6).reshape(B, 6)
            assert not arr(w, B * 4).any() and not arr(v, B * 2).any() and np.array_equal(arr(R, 4), [1., 0., 0., 1.])
            xn = om.f(x, upv[:, :2], upv[:, 2:], .5)
            arr(Xp, B * 4)[:] = xn.ravel()
            arr(Y, B * 2)[:] = om.h(xn, upv[:, :2], upv[:, 2:], .5).ravel()
            arr(q, B)[:] = 1.
            return 0
    monkeypatch.setattr(_lib, 'lib', lambda: Lib)
    rng = np.random.default_rng(0)
    X = np.array([.1, 40., .5, .2]) * (1 + .1 * rng.standard_normal((5, 4)))
    U = rng.uniform(0, .3, (5, 2))
    p = [100., 4., 1., 0.]
    xn, y = m.step(X, U, p)
    ref = om.f(X, U, np.tile(p, (5, 1)), .5)
    np.testing.assert_allclose(xn, ref, rtol=1e-14)
    np.testing.assert_allclose(y, ref[:, [0, 2]], rtol=1e-14)
    with pytest.raises(RuntimeError, match="The model has 2 inputs"):
        m.step(X, None, p)
    with pytest.raises(ValueError, match="does not match"):
        m.step(X, U[:3], p)


def test_filters_need_a_model_that_is_set_up():
    """estimator/base.py:74-76; tests/test_KFs.py:24-34 - and the order of the checks: linearity first (kf.py:340-346)."""
    from hilo_mpc_amd import EKF, KF, PF, UKF
    for cls in (EKF, UKF):
        with pytest.raises(RuntimeError, match="Model is not set up. Run Model.setup\\(\\) before passing it to the Kalman filter."):
            cls(Model('chemostat4').discretize('rk4'))
    with pytest.raises(RuntimeError, match="before passing it to the particle filter."):
        PF(Model('chemostat4').discretize('rk4'))
    with pytest.raises(ValueError, match="The supplied model is nonlinear"):
        KF(Model('toy1d'))
    with pytest.raises(RuntimeError, match="Model is not set up"):
        KF(Model('linear2').discretize('erk', order=1))


def test_soft_constraint_weights_are_used_as_the_matrix_given():
    """`e^T weight e` (modeling.py:870, :878): a matrix stays the matrix it is - round 3 found `[[50, 0], [0, 80]]` reduced to the
    vector of its diagonal on the way to the device (two of the four entries read from beyond the array)."""
    import numpy as np
    import pytest
    from hilo_mpc_amd.nmpc import _constraint_weight
    W = _constraint_weight([[50., 1.], [2., 80.]], 2, 'stage constraint')
    assert W.shape == (2, 2) and W.flags['C_CONTIGUOUS'] and W[0, 1] == 1. and W[1, 0] == 2.
    np.testing.assert_array_equal(_constraint_weight(3., 2, 'c'), 3. * np.eye(2))
    np.testing.assert_array_equal(_constraint_weight([1., 2.], 2, 'c'), np.diag([1., 2.]))
    np.testing.assert_array_equal(_constraint_weight([[7.]], 1, 'c'), [[7.]])
    with pytest.raises(ValueError, match="must be a 2 x 2 matrix"):
        _constraint_weight([[1., 2., 3.]], 2, 'terminal constraint')
