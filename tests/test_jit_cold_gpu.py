"""Run-time compilation exercised COLD on the GPU box (what the reference does at `Model.setup()` -> CasADi graph
-> `ca.nlpsol`, dynamic_model.py:1293-1553, mpc.py:1778-1787): the code-object cache points at an empty directory and every
problem below is one no other test builds (own horizon / own model), so that neither the disk cache that travels with the tree nor
the modules already loaded in this process can serve it - hiprtc must compile, a new .hsaco must appear, and the results must be
those of the oracle.  The compile times are printed (`pytest -s`) and asserted to stay below a generous bound."""
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hsaco(d):
    return sorted(f for f in os.listdir(d) if f.endswith('.hsaco'))


def test_cold_compile_nmpc_kf_smpc(tmp_path, monkeypatch):
    monkeypatch.setenv('HILO_JIT_CACHE', str(tmp_path))
    import sympy as sp
    from oracle import kf as okf
    from oracle.models import OracleModel
    from oracle.nmpc import DenseIpm
    from tests.problems import C2, c2_x0, oracle_problem, product_nmpc, symbolic_model
    times = {}
    # ---- NMPC: tracking policy on the chemostat written as expressions, horizon 7 (no other test uses it) ----
    spec = dict(C2, N=7)
    assert _hsaco(tmp_path) == []
    t0 = time.perf_counter()
    nmpc = product_nmpc(spec, model=symbolic_model('chemostat4'))
    times['nmpc'] = time.perf_counter() - t0
    assert nmpc._jit and len(_hsaco(tmp_path)) == 1
    x0 = c2_x0(6)
    ref = DenseIpm(oracle_problem(spec)).solve(x0, spec['p'])
    u0 = nmpc.optimize(x0, cp=spec['p'])
    assert np.array_equal(nmpc.solver_status_code, ref['status']) and np.all(ref['status'] == 1)
    np.testing.assert_allclose(u0, ref['u0'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy(), ref['f'], rtol=1e-8)
    # the same problem again: served from the cache just written / the loaded module, no further code object
    t0 = time.perf_counter()
    again = product_nmpc(spec, model=symbolic_model('chemostat4'))
    times['nmpc_cached'] = time.perf_counter() - t0
    assert len(_hsaco(tmp_path)) == 1
    np.testing.assert_array_equal(again.optimize(x0, cp=spec['p']), u0)

    # ---- Kalman filter: a two-state model of its own ----
    from hilo_mpc_amd import EKF, Model
    m = Model(name='cold2')
    m.set_equations(equations='''
    da/dt = -0.31*a(t) + v(k)
    db/dt = 0.31*a(t) - 0.17*b(t)^2
    y(k) = b(t) + 0.05*a(t)^2
    ''')
    a, b, v = sp.symbols('a b v')
    om = OracleModel('cold2', -1, [a, b], [v], [], [-0.31 * a + v, 0.31 * a - 0.17 * b ** 2], [b + 0.05 * a ** 2])
    md = m.discretize('rk4').setup(dt=.5)
    f = EKF(md)
    t0 = time.perf_counter()
    f.setup()
    times['kf'] = time.perf_counter() - t0
    assert len(_hsaco(tmp_path)) == 2
    rng = np.random.default_rng(2)
    B = 32
    x = np.array([1., .5]) * (1 + .2 * rng.uniform(-1, 1, (B, 2)))
    P = np.tile(.1 * np.eye(2), (B, 1, 1))
    u = rng.uniform(0, .3, (B, 1))
    y = x[:, 1:2] + .01 * rng.normal(size=(B, 1))
    f.Q, f.R = 1e-4, 1e-2
    f.set_initial_guess(x, P0=P)
    f.estimate(y=y, u=u)
    want, _ = okf.kf_step(om.discretize(4), okf.pack(x, P), y, u, np.zeros((B, 0)), 1e-4, 1e-2, .5)
    np.testing.assert_allclose(f.x.cpu().numpy(), want[:, :, 0], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(f.P.cpu().numpy(), want[:, :, 1:], rtol=1e-9, atol=1e-13)

    # ---- stochastic NMPC: the surrogate with a learned term, horizon 9 (the fixture's points do not depend on the horizon) ----
    import json
    from tests.problems import SMPC_CASES, smpc_models, smpc_product, smpc_product_gp
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'smpc.json')))['siso']
    mm, _ = smpc_models('siso')
    gp = smpc_product_gp(mm.dynamical_state_names[SMPC_CASES['siso']['features'][0]])
    case = dict(SMPC_CASES['siso'], N=9)
    monkeypatch.setitem(SMPC_CASES, 'siso', case)
    t0 = time.perf_counter()
    smpc = smpc_product('siso', gp)
    times['smpc'] = time.perf_counter() - t0
    assert smpc._jit and len(_hsaco(tmp_path)) == 3
    pts = gold['points']
    xa, uu, pp, ff = (np.array([q[k] for q in pts]) for k in ('xa', 'u', 'p', 'f'))
    np.testing.assert_allclose(smpc.plant_step(xa, uu, cp=pp).cpu().numpy(), ff, rtol=1e-10, atol=1e-12)
    print('hiprtc seconds (cold):', {k: round(v, 2) for k, v in times.items()})
    assert times['nmpc'] > 5 * times['nmpc_cached']          # the first setup really compiled
    assert max(times.values()) < 120.
