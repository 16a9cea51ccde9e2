"""Model descriptions: the zoo of device functors (hilo_mpc_amd/csrc/hilo_models.h) and models written as expressions
(`set_dynamical_states / set_inputs / set_parameters / set_algebraic_states`, `set_dynamical_equations`,
`set_algebraic_equations`, `set_measurement_equations`, `substitute_from`), which hilo_mpc_amd/codegen.py turns into a functor of
the same shape for the run-time compiler.

The reference's `Model` (hilo_mpc/modules/dynamic_model/dynamic_model.py) is a symbolic CasADi container; the
hot path only needs its *description*: dimensions, whether it is discrete, the discretisation recipe and the
sampling interval.  Method names follow the reference (`discretize`, `setup`, `is_linear`).
"""
import copy

import numpy as np

from . import expr as _expr
from .expr import Expr, SymVector


# name -> (model id, is_linear, state names, input names, parameter names, measurement names)
ZOO = {
    'lti': (0, True, None, None, None, None),
    'toy1d': (1, False, ['x'], [], [], ['y']),
    'bioreactor3': (2, False, ['T', 'cB', 'cS'], ['D'], ['alpha', 'T_amb', 'mu_0', 'mu_1', 'K', 'Y'],
                    ['y_0', 'y_1']),
    'chemostat4': (3, False, ['X', 'S', 'P', 'I'], ['DS', 'DI'], ['Sf', 'If', 'ISF', 'IRF'], ['yX', 'yP']),
    'pendulum4': (4, False, ['x', 'v', 'theta', 'omega'], ['F'], [], ['yx', 'yv', 'ytheta', 'tomega']),
    'robot6': (5, False, ['px', 'vx', 'py', 'vy', 'psi', 'omega'], ['a', 'alpha'], [], ['ypx', 'ypy']),
    'cstr3': (6, False, ['C_A', 'C_B', 'T'], ['Q'], [], ['r']),           # CSTR_Example.ipynb cell 6
    'linear2': (7, True, ['x_1', 'x_2'], ['u'], ['k_1', 'k_2'], ['y']),
    'chemostat4_gp': (8, False, ['X', 'S', 'P', 'I'], ['DS', 'DI'], ['Sf', 'If', 'ISF', 'IRF'], ['yX', 'yP']),
}
# zoo models whose right-hand side has a slot for a learned term: base name -> (label, features, hybrid functor)
LEARNABLE = {'chemostat4': ('mu', ['S', 'I'], 'chemostat4_gp')}
NATIVE_DISCRETE = {'toy1d', 'lti'}
# device functor of each zoo model (csrc/hilo_models.h): what a run-time compiled problem aliases as `UserModel`
ZOO_FUNCTOR = {'toy1d': 'Toy1D', 'bioreactor3': 'Bioreactor3', 'chemostat4': 'Chemostat4', 'pendulum4': 'Pendulum4',
               'robot6': 'Robot6', 'linear2': 'Linear2', 'cstr3': 'Cstr3'}
MODEL_USER = 100    # HILO_MODEL_USER: the model is defined by expressions and compiled at setup (csrc/hilo_jit.hip)


class Model:
    """Zoo model: `Model('chemostat4').discretize('rk4').setup(dt=1.)`, or a model defined by expressions like the reference's
    (`Model.set_dynamical_states / set_inputs / set_parameters / set_dynamical_equations / set_measurement_equations`,
    hilo_mpc/modules/dynamic_model/dynamic_model.py:1293-1553):

        m = Model(name='plant')
        x = m.set_dynamical_states(['C_A', 'C_B', 'T']); u = m.set_inputs(['Q'])
        r = 5000 * exp(-1e4 / (1.987 * x[2])) * x[0] - 1e6 * exp(-1.5e4 / (1.987 * x[2])) * x[1]
        m.set_dynamical_equations([(1 - x[0]) / 60 - r, -x[1] / 60 + r, 5 * r + (400 - x[2]) / 60 + u[0] / 1e5])
        m.setup(dt=1.)

    Such a model is emitted as a device functor (hilo_mpc_amd/codegen.py) and compiled with hiprtc at `NMPC.setup()`.

    For `Model('lti', A=..., B=..., C=...)` the matrices define a discrete LTI system
    (`x+ = A x + B u`, `y = C x`, cf. tests/test_LMPC.py:8-19)."""

    def __init__(self, name=None, A=None, B=None, C=None, discrete=None, id=None, plot_backend=None, **kwargs):
        if name == 'chemostat4_gp':
            raise ValueError("build the hybrid model with Model('chemostat4').substitute_from(gp)")
        if name not in ZOO:
            # a model defined by expressions
            self.name = name
            self._symbolic = True
            self.model_id, self._linear = MODEL_USER, False
            self.dt, self.erk_order, self.n_sub, self._is_setup = None, 0, 1, False
            self._native_discrete = bool(discrete)
            self.dynamical_state_names, self.input_names, self.parameter_names, self.measurement_names = [], [], [], []
            self.n_x = self.n_u = self.n_p = self.n_y = 0
            self._ode, self._meas = None, []
            self.learned = None
            self._gps = []              # trained GPs substituted into the equations (substitute_from)
            self.algebraic_state_names, self.n_z, self._alg = [], 0, []    # semi-explicit DAE: 0 = g(x, z, u, p)
            return
        self._symbolic = False
        self.name = name
        self.model_id, self._linear, xs, us, ps, ys = ZOO[name]
        self.dt = None
        self.erk_order = 0          # 0: not discretised
        self.n_sub = 1
        self._is_setup = False
        if name == 'lti':
            A = np.atleast_2d(np.asarray(A, dtype=float))
            B = np.atleast_2d(np.asarray(B, dtype=float))
            if B.shape[0] != A.shape[0]:
                B = B.T
            C = np.eye(A.shape[0]) if C is None else np.atleast_2d(np.asarray(C, dtype=float))
            self.A, self.B, self.C = A, B, C
            self.n_x, self.n_u, self.n_y = A.shape[0], B.shape[1], C.shape[0]
            self.n_p = A.size + B.size + C.size
            self._native_discrete = True
            xs = [f'x_{i}' for i in range(self.n_x)]
            us = [f'u_{i}' for i in range(self.n_u)]
            ps, ys = [], [f'y_{i}' for i in range(self.n_y)]
        else:
            # dimensions are those of the device functor; tests/test_abi.py checks this table against
            # hilo_model_dims() of the built library
            self.n_x, self.n_u, self.n_p, self.n_y = len(xs), len(us), len(ps), len(ys)
            self._native_discrete = name in NATIVE_DISCRETE if discrete is None else bool(discrete)
        self.dynamical_state_names, self.input_names = list(xs), list(us)
        self.parameter_names, self.measurement_names = list(ps), list(ys)
        self.learned = None         # trained GaussianProcess substituted into the right-hand side

    # -- reference-like surface ---------------------------------------------------------------
    # symbols for expressions (the reference's `model.x`, `model.u`, `model.p` SX vectors)
    x = property(lambda s: SymVector('x', s.dynamical_state_names))
    u = property(lambda s: SymVector('u', s.input_names))
    p = property(lambda s: SymVector('p', s.parameter_names))
    z = property(lambda s: SymVector('z', getattr(s, 'algebraic_state_names', [])))

    @property
    def discrete(self):
        return self._native_discrete or self.erk_order > 0

    # -- models defined by expressions ----------------------------------------------------------------
    def _declare(self, attr, names, kind):
        if not self._symbolic:
            raise RuntimeError(f"'{self.name}' is a model of the device zoo; its variables are fixed")
        names = [names] if isinstance(names, str) else list(names)
        if len(set(names)) != len(names):
            raise ValueError(f"duplicate names in {names}")
        setattr(self, attr, names)
        self.n_x, self.n_u, self.n_p = len(self.dynamical_state_names), len(self.input_names), len(self.parameter_names)
        self.n_z = len(self.algebraic_state_names)
        return SymVector(kind, names)

    def set_dynamical_states(self, *names):
        """dynamic_model.py `set_dynamical_states`: returns the symbols of the states."""
        return self._declare('dynamical_state_names', names[0] if len(names) == 1 and not isinstance(names[0], str) else names, 'x')

    def set_inputs(self, *names):
        return self._declare('input_names', names[0] if len(names) == 1 and not isinstance(names[0], str) else names, 'u')

    def set_parameters(self, *names):
        return self._declare('parameter_names', names[0] if len(names) == 1 and not isinstance(names[0], str) else names, 'p')

    def set_measurements(self, *names):
        """dynamic_model.py `set_measurements`: names of the measurements (their equations follow with
        `set_measurement_equations`)."""
        if not self._symbolic:
            raise RuntimeError(f"'{self.name}' is a model of the device zoo; its variables are fixed")
        names = names[0] if len(names) == 1 and not isinstance(names[0], str) else names
        names = [names] if isinstance(names, str) else list(names)
        if len(set(names)) != len(names):
            raise ValueError(f"duplicate names in {names}")
        self._declared_measurements = names
        return SymVector('y', names)

    def set_algebraic_states(self, *names):
        """dynamic_model.py `set_algebraic_states`: the z of a semi-explicit DAE  dx/dt = f(x, z, u, p),  0 = g(x, z, u, p)."""
        return self._declare('algebraic_state_names', names[0] if len(names) == 1 and not isinstance(names[0], str) else names, 'z')

    def set_algebraic_equations(self, equations):
        """dynamic_model.py `set_algebraic_equations`: one residual per algebraic state (index 1: dg/dz regular)."""
        if not self._symbolic:
            raise RuntimeError(f"'{self.name}' is a model of the device zoo; its equations are fixed")
        eqs = self._parse(equations)
        if len(eqs) != self.n_z:
            raise ValueError(f"the model has {self.n_z} algebraic states but {len(eqs)} algebraic equations were supplied")
        self._alg = eqs

    def _parse(self, eqs):
        """Expressions, or strings of the model's variable names with the functions of the reference's table (the right-hand side of
        an optional `... = ` is taken, like the reference's equation strings, util/parsing.py)."""
        from .parsing import FUNCTIONS
        ns = dict(FUNCTIONS)                        # the reference's function table (util/parsing.py:36-58)
        ns['dt'] = Expr('dt', name='dt')            # the sampling interval inside the equations of a discrete model
        for vec in (self.x, self.u, self.p, self.z):
            ns.update({n: vec[n] for n in vec._names})
        out = []
        for e in ([eqs] if isinstance(eqs, (Expr, str, int, float)) else list(eqs)):
            if isinstance(e, str):
                e = eval(e.split('=')[-1].replace('^', '**'), {'__builtins__': {}}, dict(ns))
            out.append(Expr.wrap(e))
        return out

    def set_dynamical_equations(self, equations):
        """dynamic_model.py:1293-1405: one right-hand side per state (dx/dt for a continuous model, x+ for a discrete one)."""
        if not self._symbolic:
            raise RuntimeError(f"'{self.name}' is a model of the device zoo; its equations are fixed")
        eqs = self._parse(equations)
        if len(eqs) != self.n_x:
            raise ValueError(f"the model has {self.n_x} dynamical states but {len(eqs)} equations were supplied")
        self._ode = eqs
        self._linear = self._check_linearity()

    def set_measurement_equations(self, equations):
        """dynamic_model.py:1407-1460; the measurements carry the names given to `set_measurements`, y_0, y_1, ... otherwise (the
        reference's defaults)."""
        if not self._symbolic:
            raise RuntimeError(f"'{self.name}' is a model of the device zoo; its equations are fixed")
        self._meas = self._parse(equations)
        named = getattr(self, '_declared_measurements', None)
        self.measurement_names = list(named) if named and len(named) == len(self._meas) else [f'y_{i}' for i in range(len(self._meas))]
        self.n_y = len(self._meas)
        self._linear = self._check_linearity()

    def set_equations(self, equations=None, ode=None, meas=None, alg=None, **kwargs):
        """dynamic_model.py:1508-1553: `equations` as text (one string with line breaks, or a list of lines; grammar in
        hilo_mpc_amd/parsing.py) declares states, measurements, algebraic states, inputs and parameters by the way they appear;
        or the equation groups separately (`ode=`, `alg=`, `meas=`)."""
        if equations is not None:
            if not self._symbolic:
                raise RuntimeError(f"'{self.name}' is a model of the device zoo; its equations are fixed")
            if not (isinstance(equations, str) or (isinstance(equations, (list, tuple)) and all(isinstance(q, str) for q in equations))):
                raise TypeError("equations must be a string or a list of strings")
            from .parsing import parse_dynamic_equations
            pm = parse_dynamic_equations(equations, discrete=self._native_discrete, x=self.dynamical_state_names,
                                         y=self.measurement_names if self._meas else (), z=self.algebraic_state_names,
                                         u=self.input_names, p=self.parameter_names)
            if any(e is None for e in pm.ode):
                missing = [n for n, e in zip(pm.x, pm.ode) if e is None]
                raise ValueError(f"no dynamical equation for the state(s) {missing}")
            self.dynamical_state_names, self.input_names, self.parameter_names = pm.x, pm.u, pm.p
            self.algebraic_state_names = pm.z
            self.n_x, self.n_u, self.n_p, self.n_z = len(pm.x), len(pm.u), len(pm.p), len(pm.z)
            if len(pm.alg) != len(pm.z):
                raise ValueError(f"the model has {len(pm.z)} algebraic states but {len(pm.alg)} algebraic equations were supplied")
            self._ode, self._alg = pm.ode, pm.alg
            self._meas, self.measurement_names, self.n_y = pm.meas, list(pm.y), len(pm.meas)
            self._equation_notes = pm.notes
            self._is_setup = False
            self._linear = self._check_linearity()
            return
        if ode is not None:
            self.set_dynamical_equations(ode)
        if alg is not None:
            self.set_algebraic_equations(alg)
        if meas is not None:
            self.set_measurement_equations(meas)

    def _check_linearity(self):
        """dynamic_model.py `_check_linearity`: the right-hand sides are linear in (x, z, u) when no entry of their Jacobian depends
        on these variables (parameters may appear in the matrices)."""
        if not self._symbolic or self._ode is None:
            return self._linear
        from .expr import jacobian
        rows = list(self._ode) + list(self._alg) + list(self._meas)
        w = list(self.x) + list(self.z) + list(self.u)
        try:
            J = jacobian(rows, w)
        except NotImplementedError:
            return False
        return not any(n.op in ('x', 'z', 'u') for r in J for e in r for n in e.nodes().values())

    def user_source(self, z_guess=None, alg_at_slope=False):
        """HIP source that defines `UserModel` for the run-time compiled path: the emitted functor, or the alias of the
        zoo functor.  z_guess: start of the Newton iteration on the algebraic equations of a DAE (`set_initial_guess(z_guess=)`)."""
        from . import codegen
        if getattr(self, '_linearized', False):
            # linearize() keeps the nonlinear equations and only marks the copy (the matrices come from system_matrices()); kernels
            # compiled from it would run the NONLINEAR dynamics in absolute coordinates (the reference rewrites the equations in
            # deviation variables, dynamic_model.py:2488-2612)
            raise NotImplementedError("a linearised model (Model.linearize) is offloaded for the linear MPC and system_matrices() only; "
                                      "filters, NMPC and simulation need the model itself")
        if self._symbolic:
            if self._ode is None:
                raise RuntimeError("Model is not set up: no dynamical equations (set_dynamical_equations)")
            for e in self._ode + self._meas + self._alg:
                if e.depends_on('theta'):
                    raise ValueError("a path variable cannot appear in the model equations")
            if any(e.depends_on('dt') for e in self._ode + self._meas + self._alg):
                # the sampling interval written into the equations of a discrete model (`25*dt*x/(1 + x^2)`): a number once the
                # model is set up
                if self.dt is None:
                    raise RuntimeError("Model is not set up: the equations use the sampling interval dt (Model.setup(dt=...))")
                dtc = Expr.wrap(float(self.dt))
                n1, n2 = len(self._ode), len(self._ode) + len(self._meas)
                sub = Expr.substitute(self._ode + self._meas + self._alg, lambda n: dtc if n.op == 'dt' else None)
                m = copy.copy(self)
                m._ode, m._meas, m._alg = sub[:n1], sub[n1:n2], sub[n2:]
                return m.user_source(z_guess, alg_at_slope)
            if self.n_z and len(self._alg) != self.n_z:
                raise RuntimeError("Model is not set up: algebraic states without algebraic equations (set_algebraic_equations)")
            if self.n_z:
                if self._native_discrete:
                    raise NotImplementedError("algebraic states of a discrete model are not built")
                return codegen.dae_model_source(self.n_x, self.n_u, self.n_p, self.n_z, self._ode, self._alg, self._meas,
                                                z_guess if z_guess is not None else [0.] * self.n_z, alg_at_slope=alg_at_slope)
            helpers = [src for _, src in sorted(getattr(self, '_gp_helpers', {}).items())]
            return codegen.model_source(self.n_x, self.n_u, self.n_p, self._ode, self._meas, self._native_discrete, helpers=helpers)
        if self.name == 'lti':
            return codegen.zoo_alias(f"Lti<{self.n_x}, {self.n_u}, {self.n_y}>")
        if self.name not in ZOO_FUNCTOR:
            raise NotImplementedError(f"model '{self.name}' cannot be used in a run-time compiled problem")
        return codegen.zoo_alias(ZOO_FUNCTOR[self.name])

    def is_linear(self):
        """dynamic_model.py `is_linear`: a model of the zoo by its table; a model written as expressions when it was linearised or
        when the Jacobian of its equations with respect to states and inputs holds neither."""
        if not self._symbolic:
            return self._linear
        if getattr(self, '_linearized', False):
            return True
        if self._ode is None or self.n_z:
            return False
        from .expr import jacobian
        try:
            J = jacobian(list(self._ode) + list(self._meas), list(self.x) + list(self.u))
        except NotImplementedError:          # a node without a derivative rule (a learned term, ...): not known to be linear
            return False
        return not any(e.depends_on('x') or e.depends_on('u') for row in J for e in row)

    # ---- linearisation (dynamic_model.py:2488-2612, :3670-3684) and the matrices the linear MPC reads (mpc.py:2183-2184) -----------
    def linearize(self, name=None, trajectory=None):
        """The model linearised about its equilibrium point (`set_equilibrium_point`, default: the origin): a copy that reports
        `is_linear()` and hands out `state_matrix`, `input_matrix`, `output_matrix` - the Jacobians of the equations (of the
        explicit Runge-Kutta step when the model was discretised first, like in tests/test_LMPC.py:82-85) with respect to states
        and inputs, in deviation variables."""
        if not self._symbolic:
            if self._linear:
                print("Model is already linear. Linearization is not necessary. Nothing to be done.")
                return self
            raise NotImplementedError("the models of the device zoo carry no expressions to linearise")
        if trajectory is not None:
            raise NotImplementedError("linearisation along a trajectory is not built")
        if self._ode is None:
            print("Model is empty. Nothing to be done.")
            return self
        if getattr(self, '_linearized', False):
            print("Model is already linearized. Nothing to be done.")
            return self
        if self.is_linear():
            print("Model is already linear. Linearization is not necessary. Nothing to be done.")
            return self
        if self.n_z:
            raise NotImplementedError("linearisation of a model with algebraic states is not built")
        m = self.copy()
        if name is not None:
            m.name = name
        m._linearized = True
        m._x_eq = m._u_eq = None
        return m

    def set_equilibrium_point(self, x_eq=None, u_eq=None, z_eq=None):
        """dynamic_model.py `set_equilibrium_point`: where the Jacobians of a linearised model are taken."""
        def chk(v, n, what):
            if v is None:
                return None
            v = np.asarray(v, dtype=float).ravel()
            if v.size != n:
                raise ValueError(f"Dimension mismatch: the equilibrium point of the {what} has {v.size} entries, the model {n}")
            return v
        self._x_eq, self._u_eq = chk(x_eq, self.n_x, 'states'), chk(u_eq, self.n_u, 'inputs')

    def set_initial_parameter_values(self, p=None):
        """dynamic_model.py `set_initial_parameter_values` (here: the values the system matrices are evaluated with when none
        are handed over)."""
        p = np.asarray([] if p is None else p, dtype=float).ravel()
        if p.size != self.n_p:
            raise ValueError(f"Dimension mismatch: {p.size} parameter values for {self.n_p} parameters")
        self._p_init = p

    def _jacobian_program(self):
        """(dag, nodes of d rows / d (x, u), number of rows): rows = x+ (or dx/dt) and the measurement equations."""
        from .smpc import SMPC
        from .symdiff import Dag
        fx = SMPC._discrete_map(self, self.dt) if self.discrete else list(self._ode)
        rows = list(fx) + list(self._meas)
        if any(e.depends_on('t') for e in rows):
            raise NotImplementedError("system matrices of a model that depends on time explicitly")
        if any(e.depends_on('dt') for e in rows):      # the sampling interval written into the equations (tests/test_LMPC.py:12-13)
            if self.dt is None:
                raise RuntimeError("Model is not set up. Run Model.setup(dt=...) first: the equations hold the sampling interval.")
            rows = Expr.substitute(rows, lambda n: Expr.wrap(float(self.dt)) if n.op == 'dt' else None)
        g, memo = Dag(), {}
        rows = [g.from_expr(e, memo) for e in rows]
        w = [g.var('x', i) for i in range(self.n_x)] + [g.var('u', i) for i in range(self.n_u)]
        jac = [g.diff(r, v) for r in rows for v in w]
        return g, jac, len(rows)

    def system_matrices(self, p=None):
        """(A, B, C) of a linear (or linearised) model: numeric Jacobians of x+ = f(x, u, p) (a discrete or discretised model) or
        dx/dt = f (a continuous one that was not discretised) and of y = h(x, u, p) at the equilibrium point."""
        if not self._symbolic:
            if self.name == 'lti':
                return self.A, self.B, self.C
            raise NotImplementedError("the models of the device zoo carry no expressions to differentiate")
        if not self.is_linear():
            raise RuntimeError("The model is nonlinear: linearize it first (Model.linearize)")
        if self.discrete and not self._native_discrete and self.dt is None:
            raise RuntimeError("Model is not set up. Run Model.setup(dt=...) before asking for the matrices of a discretised model.")
        if self.n_p:
            p = getattr(self, '_p_init', None) if p is None else np.asarray(p, dtype=float).ravel()
            if p is None or p.size != self.n_p:
                raise ValueError(f"The model has {self.n_p} parameter(s): {self.parameter_names}. Their values are needed for "
                                 f"the system matrices (argument `p` / set_initial_parameter_values).")
        else:
            p = np.zeros(0)
        # the Jacobian as a straight-line program, built once per (equations, discretisation, sampling interval) and evaluated per
        # call (the linear MPC asks for one pair of matrices per stage when parameters vary along the horizon)
        # (the equation lists themselves are kept in the cache entry: an id is only unique among live objects)
        stamp = (id(self._ode), id(self._meas), self.erk_order, self.n_sub, self.dt, self.discrete)
        cache = getattr(self, '_sysmat_cache', None)
        if cache is None or cache[0] != stamp or cache[4] is not self._ode or cache[5] is not self._meas:
            g, jac, n_rows = self._jacobian_program()
            self._sysmat_cache = cache = (stamp, g, jac, n_rows, self._ode, self._meas)
        _, g, jac, n_rows = cache[:4]
        xe = getattr(self, '_x_eq', None)
        ue = getattr(self, '_u_eq', None)
        xe = np.zeros(self.n_x) if xe is None else xe
        ue = np.zeros(self.n_u) if ue is None else ue
        J = np.array(g.evaluate(jac, xe, ue, p), dtype=float).reshape(n_rows, self.n_x + self.n_u)
        nx = self.n_x
        C = J[nx:, :nx] if len(self._meas) else np.eye(nx)
        return J[:nx, :nx], J[:nx, nx:], C

    state_matrix = property(lambda s: s.system_matrices()[0])
    input_matrix = property(lambda s: s.system_matrices()[1])
    output_matrix = property(lambda s: s.system_matrices()[2])

    def lti_parameters(self):
        return np.concatenate([self.A.ravel(), self.B.ravel(), self.C.ravel()])

    def discretize(self, method='rk4', order=None, inplace=False, n_sub=1):
        """`Model.discretize` (dynamic_model.py:2113-2456): 'rk4' = classic order 4, 'erk' with order 1..4
        (modeling.py:1239-1250)."""
        if method == 'rk4':
            order = 4
        elif method == 'erk':
            order = 1 if order is None else order
        else:
            raise ValueError(f"discretisation method '{method}' is not available on the device "
                             f"(explicit Runge-Kutta 'erk'/'rk4' only)")
        if order not in (1, 2, 3, 4):
            raise NotImplementedError(f"Explicit Runge-Kutta discretization for order {order} is not yet "
                                      f"implemented.")
        m = self if inplace else copy.copy(self)
        if not m._native_discrete:
            m.erk_order = order
            m.n_sub = n_sub
        return m

    def setup(self, dt=None):
        if self._symbolic and self._ode is None:
            raise RuntimeError("Model is not set up: no dynamical equations (set_dynamical_equations)")
        if dt is not None:
            self.dt = float(dt)
        if self.dt is None:
            self.dt = 1.
        self._is_setup = True
        return self

    def substitute_from(self, obj):
        """`Model.substitute_from(gp)` (dynamic_model.py:3040-3125): the quantity named by `gp.labels` is replaced
        by the GP's posterior mean `gp.predict(features)[0]`, the features being looked up by name among the model's
        variables (:3056-3061).  The device zoo offers this for the growth rate `mu` of 'chemostat4' over the
        features (S, I) (`nmpc_hybrid_bio.ipynb`); the GP must be trained (`setup` + `fit_model` or `set_training_data`
        + `setup`) with a squared-exponential kernel and a constant/zero mean."""
        if isinstance(obj, (list, tuple, set)):
            for o in obj:
                self.substitute_from(o)
            return self
        if self._symbolic:
            return self._substitute_symbolic(obj)
        if self.name not in LEARNABLE:
            raise NotImplementedError(f"model '{self.name}' has no learnable term in the device zoo "
                                      f"(available: {sorted(LEARNABLE)})")
        label, features, hybrid = LEARNABLE[self.name]
        if list(getattr(obj, 'labels', [])) != [label] or list(getattr(obj, 'features', [])) != features:
            raise ValueError(f"model '{self.name}' takes a learned model with labels ['{label}'] and features "
                             f"{features}; got labels {getattr(obj, 'labels', None)}, features "
                             f"{getattr(obj, 'features', None)}")
        if getattr(obj, '_handle', None) is None:
            raise RuntimeError("The GP has not been set up (trained) yet. Run GaussianProcess.setup() first.")
        m = self
        m.name = hybrid
        m.model_id = ZOO[hybrid][0]
        m.learned = obj
        return m

    def _substitute_symbolic(self, gp):
        """The general case for a model written as expressions: the label must be a parameter (the reference only handles
        parameters either, dynamic_model.py:3000-3004 `self._p.remove`), which leaves the parameter vector; the features are
        states, inputs or remaining parameters, looked up by name (:3056-3061).  In the compiled right-hand side the
        parameter becomes `gp_se_mean(<packed GP>, features)` (csrc/hilo_models.h), differentiated like any other
        operation by the scalar type it is evaluated with."""
        labels, features = list(getattr(gp, 'labels', [])), list(getattr(gp, 'features', []))
        if len(labels) != 1 or not callable(getattr(gp, 'predict', None)):
            raise ValueError("substitute_from takes a learned model with exactly one label")
        if getattr(gp, '_handle', None) is None:
            raise RuntimeError("The GP has not been set up (trained) yet. Run GaussianProcess.setup() first.")
        if self._ode is None:
            raise RuntimeError("set the model equations before substituting a learned term")
        if labels[0] not in self.parameter_names:
            raise ValueError(f"label '{labels[0]}' is not a parameter of model '{self.name}' (parameters: "
                             f"{self.parameter_names})")
        if len(self._gps) >= 4:
            raise NotImplementedError("at most 4 learned terms per model")
        ip = self.parameter_names.index(labels[0])
        keep = [n for n in self.parameter_names if n != labels[0]]
        newp = SymVector('p', keep)
        feats = []
        for f in features:
            if f == labels[0]:
                raise ValueError(f"feature '{f}' is the label itself")
            for vec in (self.x, self.u, newp):
                if f in vec._names:
                    feats.append(vec[f])
                    break
            else:
                raise ValueError(f"feature '{f}' is not a state, input or parameter of model '{self.name}'")
        # the plain squared-exponential kernel has a device function of its own (gp_se_mean, also the variance and the mean's
        # Jacobian of the stochastic NMPC); any other stationary kernel / sum / product is compiled into the model source
        from .gp import is_plain_se, kernel_expr
        from . import codegen
        kern = getattr(gp, 'kernel', None)       # (an object without one is taken to the device function: the library checks its kernel)
        kprog = list(kern.program(len(features))) if kern is not None else None
        k = len(self._gps)
        if kprog is None or is_plain_se(kprog):
            node = Expr('gp', feats, value=k, name=labels[0])
        else:
            if len(features) > 8:
                raise NotImplementedError("a learned term inside a model takes at most 8 features")
            fe = [Expr('x', value=q, name=f"f{q}") for q in range(len(features))]
            te = [Expr('p', value=q, name=f"t{q}") for q in range(len(features))]
            if not hasattr(self, '_gp_helpers') or self._gp_helpers is None:
                self._gp_helpers = {}
            self._gp_helpers = dict(self._gp_helpers)
            self._gp_helpers[k] = codegen.gp_helper_source(k, len(features), kernel_expr(kprog, fe, te))
            node = Expr('gpk', feats, value=k, name=labels[0])

        def leaf(n):
            if n.op != 'p':
                return None
            i = int(n.value)
            return node if i == ip else (newp[i - 1] if i > ip else None)

        # every equation list that can hold a parameter leaf is re-indexed together (the algebraic equations too)
        alg = list(getattr(self, '_alg', None) or [])
        neq, nalg = len(self._ode), len(alg)
        out = Expr.substitute(self._ode + alg + self._meas, leaf)
        self._ode, self._meas = out[:neq], out[neq + nalg:]
        if nalg:
            self._alg = out[neq:neq + nalg]
        self.parameter_names, self.n_p = keep, len(keep)
        self._gps.append(gp)
        self._is_setup = False
        return self

    def copy(self, setup=True):
        m = copy.copy(self)
        m._sim = None              # the copy simulates its own trajectory
        if hasattr(self, '_gps'):
            m._gps = list(self._gps)    # learned terms substituted into the copy later must not appear in the original
        if getattr(self, '_gp_helpers', None):
            m._gp_helpers = dict(self._gp_helpers)
        return m

    # ---- the model as a PLANT: one sampling interval for a batch of states (dynamic_model.py:3360-3400, :3911-4000) -----------------
    def _plant_handle(self, device_index=None):
        """A filter handle of the library carries exactly what a plant step needs - the model functor (zoo, or compiled from the
        expressions), the sampling interval and the discretisation - and `hilo_pf_function` with one particle per instance and
        zero noise IS that step: x+ = Phi(x, u, p), y = h(x+, u, p) (a continuous model is integrated with eight classic
        Runge-Kutta steps per interval in place of the reference's CVODES)."""
        h = getattr(self, '_plant', None)
        if h is None or h._model_stamp != (self.dt, self.erk_order, self.n_sub, id(self._ode) if self._symbolic else None):
            from .estimator import ExtendedKalmanFilter
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                h = ExtendedKalmanFilter(self, device_index=device_index)
            h.setup()
            h._model_stamp = (self.dt, self.erk_order, self.n_sub, id(self._ode) if self._symbolic else None)
            self._plant = h
        return h

    def step(self, x, u=None, p=None, device_index=None):
        """x [B, n_x] -> (x_next [B, n_x], y [B, n_y]) after one sampling interval, on the device (numpy in -> numpy out)."""
        import torch
        from . import _lib
        from ._device import ptr, stream_ptr, to_dev
        if not self._is_setup:
            raise RuntimeError("Model is not set up. Run Model.setup() before running simulations.")
        h = self._plant_handle(device_index)
        dev = h._dev
        host = not isinstance(x, torch.Tensor)
        xt = to_dev(x, dev).reshape(-1, self.n_x).contiguous()
        B = xt.shape[0]
        parts = []
        for v, n, what in ((u, self.n_u, 'inputs'), (self.lti_parameters() if self.name == 'lti' else p, h._n_p, 'parameters')):
            if n:
                if v is None:
                    raise RuntimeError(f"The model has {n} {what}; pass them to step() / simulate().")
                t = to_dev(v, dev).reshape(-1, n)
                if t.shape[0] not in (1, B):
                    raise ValueError(f"{what}: batch {t.shape[0]} does not match {B} states")
                parts.append(t.expand(B, -1))
        up = torch.cat(parts, dim=1).contiguous() if parts else None
        ny = h._n_y
        zeros_x = torch.zeros(B, 1, self.n_x, dtype=torch.float64, device=dev)
        zeros_y = torch.zeros(B, 1, ny, dtype=torch.float64, device=dev)
        R = torch.eye(ny, dtype=torch.float64, device=dev)
        xn, y = torch.empty(B, 1, self.n_x, dtype=torch.float64, device=dev), torch.empty(B, 1, ny, dtype=torch.float64, device=dev)
        q = torch.empty(B, 1, dtype=torch.float64, device=dev)
        _lib.check(_lib.lib().hilo_pf_function(h._handle, B, 1, ptr(xt), ptr(zeros_y), ptr(up), (up.shape[1] if up is not None else 0),
                                               ptr(zeros_x), ptr(zeros_y), ptr(R), 0, ptr(xn), ptr(y), ptr(q), stream_ptr(dev)))
        xn, y = xn[:, 0], y[:, 0]
        return (xn.cpu().numpy(), y.cpu().numpy()) if host else (xn, y)

    def set_initial_conditions(self, x0, t0=0., z0=None):
        """dynamic_model.py:3360-3400 (a batch of states is allowed: [B, n_x])."""
        if not self._is_setup:
            raise RuntimeError("Model is not set up. Run Model.setup() before setting the initial conditions.")
        x = np.atleast_2d(np.asarray(x0.cpu() if hasattr(x0, 'cpu') else x0, dtype=float))
        if x.shape[1] != self.n_x:
            x = x.T
        if x.shape[1] != self.n_x:
            raise ValueError(f"Dimension mismatch. Supplied dimension for the initial states is {x.shape[1]}, but required "
                             f"dimension is {self.n_x}.")
        self._sim = {'t': [float(t0)], 'x': [x], 'y': [], 'u': []}

    def simulate(self, u=None, p=None, steps=1, **kwargs):
        """dynamic_model.py:3911-4000: advance the stored state by `steps` sampling intervals with the inputs held; results in
        `model.solution` (`solution['x:f']` the last state, `solution['x']` the trajectory)."""
        if not self._is_setup:
            raise RuntimeError("Model is not set up. Run Model.setup() before running simulations.")
        if getattr(self, '_sim', None) is None:
            raise RuntimeError("No initial dynamical states found. Please set initial conditions before simulating the model.")
        if u is not None:
            u = np.asarray(u.cpu() if hasattr(u, 'cpu') else u, dtype=float)
            u = u.reshape(1, -1) if u.size == self.n_u else (u.T if u.ndim == 2 and u.shape[0] == self.n_u and u.shape[1] != self.n_u else u)
        for _ in range(int(steps)):
            x, y = self.step(self._sim['x'][-1], u, p)
            self._sim['x'].append(x), self._sim['y'].append(y), self._sim['u'].append(u)
            self._sim['t'].append(self._sim['t'][-1] + self.dt)

    @property
    def solution(self):
        """`model.solution['x:f']`, `['y:f']`, `['t:f']`, `['x']` ... (the key language of base.py:2426-2470 for the last value /
        the whole series; single instance: column vectors like the reference's)."""
        sim = getattr(self, '_sim', None)

        class _View:
            def __getitem__(_, key):
                if sim is None:
                    raise KeyError(key)
                name, _, sel = key.partition(':')
                series = sim[name]
                if not series:
                    raise KeyError(key)
                if sel in ('f', '-1'):
                    v = np.asarray(series[-1])
                    return v.T if (v.ndim == 2 and v.shape[0] == 1) else v
                if sel == '0':
                    v = np.asarray(series[0])
                    return v.T if (v.ndim == 2 and v.shape[0] == 1) else v
                a = np.asarray(series)
                return a[:, 0].T if (a.ndim == 3 and a.shape[1] == 1) else a

            get_by_id = __getitem__
        return _View()
