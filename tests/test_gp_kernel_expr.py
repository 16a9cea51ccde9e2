"""CPU: the kernel of a learned term as an expression tree (hilo_mpc_amd/gp.py::kernel_expr - what is compiled into a run-time
compiled model when the kernel is not the plain squared exponential) against the oracle's kernels, which are pinned by the
reference's known answers (tests/test_oracle_gp.py); and the emitted helper compiled by hiprtc without a GPU."""
import numpy as np
import pytest

from hilo_mpc_amd import gp as G
from hilo_mpc_amd.expr import Expr
from oracle import gp as ogp
from tests.problems import eval_exprs

KW = dict(active_dims=[0, 1], length_scales=[10., 1.5], ard=True)
CASES = {
    'matern_32': (lambda: G.Kernel.matern_32(**KW, signal_variance=.7), {'type': 'matern_32', 'kwargs': dict(KW, signal_variance=.7)}),
    'matern_52': (lambda: G.Kernel.matern_52(**KW), {'type': 'matern_52', 'kwargs': KW}),
    'exponential': (lambda: G.Kernel.exponential(**KW), {'type': 'exponential', 'kwargs': KW}),
    'rational_quadratic': (lambda: G.Kernel.rational_quadratic(**KW, alpha=1.7), {'type': 'rational_quadratic', 'kwargs': dict(KW, alpha=1.7)}),
    'sum': (lambda: G.Kernel.squared_exponential(**KW) + G.Kernel.rational_quadratic(active_dims=[1], length_scales=2., alpha=.8),
            {'type': 'sum', 'children': [{'type': 'squared_exponential', 'kwargs': KW},
                                         {'type': 'rational_quadratic', 'kwargs': dict(active_dims=[1], length_scales=2., alpha=.8)}]}),
    'product': (lambda: G.Kernel.matern_52(**KW) * G.Kernel.constant(bias=1.3),
                {'type': 'product', 'children': [{'type': 'matern_52', 'kwargs': KW}, {'type': 'constant', 'kwargs': dict(bias=1.3)}]}),
}


def _leaves(nf):
    return [Expr('x', value=q, name=f'f{q}') for q in range(nf)], [Expr('p', value=q, name=f't{q}') for q in range(nf)]


@pytest.mark.parametrize('name', sorted(CASES))
def test_kernel_expression_equals_the_oracle_kernel(name):
    make, spec = CASES[name]
    prog = make().program(2)
    assert not G.is_plain_se(prog)
    f, t = _leaves(2)
    ke = G.kernel_expr(prog, f, t)
    rng = np.random.default_rng(3)
    X, Y = rng.uniform(0, 40, (2, 6)), rng.uniform(0, 40, (2, 5))
    Y[:, 0] = X[:, 0]                                        # a query on a training point: the value is still exact
    ref = ogp.kernel(spec, X, Y)
    val = np.array([[eval_exprs([ke], X[:, i], [], Y[:, j])[0] for j in range(5)] for i in range(6)])
    np.testing.assert_allclose(val, ref, rtol=1e-13, atol=1e-15)


def test_plain_squared_exponential_keeps_its_device_function():
    assert G.is_plain_se(G.Kernel.squared_exponential(**KW).program(2))
    assert G.is_plain_se(G.Kernel.squared_exponential(active_dims=[0], length_scales=2.).program(1))
    assert not G.is_plain_se((G.Kernel.squared_exponential(**KW) + G.Kernel.constant(bias=1.)).program(2))


def test_kernels_without_an_expression_are_refused():
    f, t = _leaves(1)
    with pytest.raises(NotImplementedError, match="squared-exponential / gamma-exponential"):
        G.kernel_expr(G.Kernel.periodic(active_dims=[0]).program(1), f, t)


class _TrainedStub:
    """What `Model.substitute_from` looks at, without a device: features, labels, kernel, a handle that is not None."""
    def __init__(self, kernel, features, labels):
        self.kernel, self.features, self.labels, self._handle = kernel, features, labels, object()

    def predict(self, *a):                      # pragma: no cover
        raise RuntimeError("stub")


def test_model_with_a_matern_learned_term_compiles_here(tmp_path, monkeypatch):
    """The emitted helper inside the model source, compiled by hiprtc for every scalar type the engine evaluates the model with
    (values, duals, second-order Taylor numbers) - no GPU needed (HILO_JIT_COMPILE_ONLY)."""
    from tests.problems import C4, product_nmpc, symbolic_model
    monkeypatch.setenv('HILO_JIT_COMPILE_ONLY', '1')
    monkeypatch.setenv('HILO_JIT_CACHE', str(tmp_path))
    m = symbolic_model('chemostat4_mu')
    m.substitute_from(_TrainedStub(G.Kernel.matern_52(**KW) + G.Kernel.rational_quadratic(active_dims=[1], length_scales=2.),
                                   ['S', 'I'], ['mu']))
    assert m.parameter_names == ['Sf', 'If', 'ISF', 'IRF']
    src = m.user_source()
    assert 'hilo_user_gpk0(const double* g, const T* f)' in src and 'hilo_user_gpk0(hilo_user_gp[0]' in src and 'gp_se_mean' not in src
    product_nmpc(C4, model=m)
    assert any(f.endswith('.hsaco') for f in __import__('os').listdir(tmp_path))
