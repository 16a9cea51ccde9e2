"""Model equations given as TEXT - what `Model.set_equations(equations=...)` of the reference accepts
(hilo_mpc/modules/dynamic_model/dynamic_model.py:291-380, :1508-1553; grammar of hilo_mpc/util/parsing.py:246-545):

    dx_1/dt = -k_1*x_1(t) + u(k)          d/dt(x(t)) = ...        differential equation  -> state x_1
    x(k+1) = x(k)/2 + 25*dt*x(k)/(1 + x(k)^2)                      difference equation (discrete models)
    0 = z(t) - (v(t)^2 + w(t)^2)          z(t) = v(t)^2 + w(t)^2   algebraic equation (implicit / explicit form) -> z
    y(k) = x_2(t)                                                  measurement equation   -> measurement y
    g = 9.81                                                       constant
    r = k_0*exp(-E/T(t))*c(t)                                      auxiliary definition, substituted where it is used
    # comment              name | description: ...                 ignored / annotation (kept as text)
    a long right-hand side ...                                     continued on the next line

Variables that are not declared elsewhere are recognised by their time argument: `name(t)` on a right-hand side is an algebraic
state, `name(k)` an input (piecewise constant), every other free identifier a parameter - in the order of their first
appearance (differential equations first, then algebraic, then measurement equations, like the reference processes them).
`^` is the power operator.  The result is expression trees (hilo_mpc_amd/expr.py), evaluated with Python's own parser on a
namespace of the model's symbols.
"""
import ast
import re

from . import expr as _expr
from .expr import Expr

# the reference's table (util/parsing.py:36-58), name for name
FUNCTIONS = {n: getattr(_expr, n) for n in ('sqrt', 'exp', 'log', 'log10', 'sign', 'sin', 'cos', 'tan', 'arcsin', 'arccos',
                                             'arctan', 'arctan2', 'sinh', 'cosh', 'tanh', 'arsinh', 'arcosh', 'artanh')}
FUNCTIONS.update(abs=_expr.fabs, min=_expr.fmin, max=_expr.fmax)
_TIMED = re.compile(r'([A-Za-z_][A-Za-z0-9_]*)\(([kt0-9+\-]{1,3})\)')


class ParsedModel:
    def __init__(self):
        self.x, self.y, self.z, self.u, self.p = [], [], [], [], []
        self.ode, self.alg, self.meas = [], [], []
        self.const, self.notes = {}, {}


def _logical_lines(equations):
    """Blanks removed, `...` continuations joined, comments and lines without '=' dropped, annotations split off."""
    lines = equations.split('\n') if isinstance(equations, str) else list(equations)
    out, notes, pending = [], {}, ''
    for raw in lines:
        if '|' in raw:
            var, prop = raw.split('|', 1)
            if ':' in prop:
                key, val = prop.split(':', 1)
                notes.setdefault(var.strip(), {})[key.strip()] = val.strip()
            continue
        text = pending + raw.replace(' ', '').replace('\t', '')
        if text.endswith('...'):
            pending = text[:-3]
            continue
        pending = ''
        if not text or text.startswith('#') or '=' not in text:
            continue
        out.append(text.replace('^', '**'))
    return out, notes


def parse_dynamic_equations(equations, discrete=False, x=(), y=(), z=(), u=(), p=()):
    """Returns a ParsedModel: names of states / measurements / algebraic states / inputs / parameters (the given ones first,
    discovered ones appended) and the equations as expression trees on symbols ('x', i), ('z', i), ('u', i), ('p', i)."""
    m = ParsedModel()
    m.x, m.y, m.z, m.u, m.p = list(x), list(y), list(z), list(u), list(p)
    lines, m.notes = _logical_lines(equations)
    odes, algs, meas, aux = [], [], [], {}

    def strip_time(name):
        return re.sub(r'\(.*?\)', '', name)

    for text in lines:
        lhs, rhs = text.split('=', 1)
        if not discrete:
            if lhs == 'int':
                raise NotImplementedError("quadrature functions (int = ...) are not offloaded")
            hit = re.fullmatch(r'd(.+)/dt', lhs) or re.fullmatch(r'd/dt\((.+)\)', lhs)
            if hit:
                name = strip_time(hit.group(1))
                if name not in m.x:
                    m.x.append(name)
                odes.append((name, rhs))
                continue
        else:
            if lhs == 'sum':
                raise NotImplementedError("quadrature functions (sum = ...) are not offloaded")
            hit = re.fullmatch(r'(.*)\(k\+1\)', lhs)
            if hit:
                name = hit.group(1)
                if name not in m.x:
                    m.x.append(name)
                odes.append((name, rhs))
                continue
        if lhs == '0':
            algs.append(rhs)
            continue
        try:
            m.const[lhs] = float(rhs)
            continue
        except ValueError:
            pass
        hit = re.fullmatch(r'(.*)\(t\)', lhs) if not discrete else None
        if hit:                                            # explicit algebraic variable  z(t) = ...
            if hit.group(1) not in m.z:
                m.z.append(hit.group(1))
            algs.append(f"({rhs})-({hit.group(1)})")
            continue
        hit = re.fullmatch(r'(.*)\(k\)', lhs)
        if hit:                                            # measurement  y(k) = ...
            if hit.group(1) not in m.y:
                m.y.append(hit.group(1))
            meas.append((hit.group(1), rhs))
            continue
        aux[lhs] = rhs                                     # auxiliary definition

    reserved = set(FUNCTIONS) | {'dt', 't'}

    def discover(text):
        """Unknown `name(t)` -> algebraic state, unknown `name(k...)` -> input (continuous models only: in a discrete model the two
        are indistinguishable and must be declared), then the time arguments are dropped."""
        for name, arg in _TIMED.findall(text):
            if name in reserved or name in aux or name in m.const:
                continue
            if name in m.x or name in m.y or name in m.z or name in m.u or name in m.p:
                continue
            if not discrete:
                (m.z if arg == 't' else m.u).append(name)
        return _TIMED.sub(lambda q: q.group(0) if q.group(1) in FUNCTIONS else q.group(1), text)

    building = []

    def evaluate(text):
        text = discover(text)
        tree = ast.parse(text, mode='eval')
        names = []
        for node in ast.walk(tree):
            if isinstance(node, ast.Name) and node.id not in names:
                names.append(node.id)
        ns = dict(FUNCTIONS)
        for name in names:
            if name in FUNCTIONS:
                continue
            if name in aux:
                if name in building:
                    raise ValueError(f"the auxiliary definition of '{name}' refers to itself")
                if not isinstance(aux[name], Expr):
                    building.append(name)
                    aux[name] = Expr.wrap(evaluate(aux[name]))
                    building.pop()
                ns[name] = aux[name]
            elif name in m.const:
                ns[name] = m.const[name]
            elif name == 'dt':
                ns[name] = symbols['dt']
            elif name == 't':
                raise NotImplementedError("explicitly time-dependent model equations are not offloaded")
            else:
                for kind, pool in (('x', m.x), ('z', m.z), ('u', m.u), ('p', m.p)):
                    if name in pool:
                        break
                else:
                    if name in m.y:
                        raise ValueError(f"the measurement '{name}' cannot appear on a right-hand side")
                    m.p.append(name)                       # a free identifier is a parameter
                    kind, pool = 'p', m.p
                ns[name] = symbols[(kind, name)]
        return eval(compile(tree, '<equation>', 'eval'), {'__builtins__': {}}, ns)

    class _Symbols(dict):
        def __missing__(self, key):
            if key == 'dt':
                v = Expr('dt', name='dt')
            else:
                kind, name = key
                pool = {'x': m.x, 'z': m.z, 'u': m.u, 'p': m.p}[kind]
                v = Expr(kind, value=pool.index(name), name=name)
            self[key] = v
            return v
    symbols = _Symbols()

    order = {n: i for i, n in enumerate(m.x)}
    ode = [None] * len(m.x)
    for name, rhs in odes:
        ode[order[name]] = Expr.wrap(evaluate(rhs))
    m.alg = [Expr.wrap(evaluate(r)) for r in algs]
    ym = dict((name, Expr.wrap(evaluate(rhs))) for name, rhs in meas)
    m.ode = ode
    m.meas = [ym[n] for n in m.y if n in ym]
    m.y = [n for n in m.y if n in ym]
    return m
