"""GPU parity: hilo_nmpc_solve (through the reference-style NMPC class and the C ABI) vs the oracle's dense
interior-point solver on the same transcription.  Stated floating-point tolerance: both solvers stop at a scaled
KKT error of 1e-8, so primal solutions agree to ~1e-6 relative (the BASELINE.json target); status codes and the
integer index bookkeeping are compared exactly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle.nmpc import DenseIpm                                      # noqa: E402
from tests.problems import C2, c2_x0, oracle_problem, product_nmpc    # noqa: E402


def test_index_bookkeeping_bit_exact():
    nmpc = product_nmpc(C2)
    pb = oracle_problem(C2)
    assert nmpc._x_ind == pb.x_ind and nmpc._u_ind == pb.u_ind        # mpc.py:1464-1485
    assert (nmpc._n_v, nmpc._n_g) == (pb.n_v, pb.n_g) == (124, 80)


def test_c2_cold_and_warm_vs_oracle():
    B = 16
    x0 = c2_x0(B)
    pb = oracle_problem(C2)
    ipm = DenseIpm(pb)
    ref = ipm.solve(x0, C2['p'])
    assert np.all(ref['status'] == 1)
    nmpc = product_nmpc(C2)
    u = nmpc.optimize(x0, cp=C2['p'])
    st = nmpc.stats()
    assert np.array_equal(nmpc.solver_status_code, ref['status'])     # bit-exact status codes
    assert np.all(st['kkt_error'] <= 1e-8)
    v = nmpc._nlp_solution['x'].cpu().numpy()
    vr = ipm.to_v(ref)
    scale = np.maximum(1., np.abs(vr))
    assert np.max(np.abs(v - vr) / scale) < 1e-6
    np.testing.assert_allclose(u, ref['u0'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(nmpc._nlp_solution['f'].cpu().numpy(), ref['f'], rtol=1e-8)
    # multipliers of the shooting constraints (sign convention L = f + lam^T g, terminal term on Phi_{N-1})
    lam_ref = ref['lam'].copy()
    lam_ref[:, -pb.nx:] += 2 * (ref['X'][:, -1] - pb.xrefN) @ pb.WN
    np.testing.assert_allclose(nmpc._nlp_solution['lam_g'].cpu().numpy(), lam_ref, rtol=1e-5, atol=1e-6)
    # iteration counts are a property of the algorithm, not of parity; they must be of the same order
    assert abs(int(st['iter_count'].mean()) - int(ref['iters'].mean())) <= 10
    # closed loop step 2: warm start from the previous solution (mpc.py:725-726), un-shifted
    x1 = nmpc.plant_step(x0, ref['u0'], cp=C2['p']).cpu().numpy()
    x1_ref = pb.phi(x0 / pb.sx, ref['U'][:, 0], C2['p']) * pb.sx
    np.testing.assert_allclose(x1, x1_ref, rtol=1e-12, atol=1e-14)
    ref2 = ipm.solve(x1_ref, C2['p'], w0=ref['w'])
    u2 = nmpc.optimize(x1_ref, cp=C2['p'])
    assert np.array_equal(nmpc.solver_status_code, ref2['status'])
    np.testing.assert_allclose(u2, ref2['u0'], rtol=1e-6, atol=1e-7)
    assert nmpc.stats()['iter_count'].mean() < st['iter_count'].mean()   # warm start pays


def test_single_instance_shapes_and_errors():
    nmpc = product_nmpc(C2)
    u = nmpc.optimize([.1, 40., 0., 0.], cp=C2['p'])
    assert u.shape == (2, 1)
    xp, up, _ = nmpc.return_prediction()
    assert xp.shape == (1, 4, 21) and up.shape == (1, 2, 20)
    with pytest.raises(ValueError, match="We have an issue mate, the x0 you supplied has dimension 3"):
        nmpc.optimize([.1, 40., 0.], cp=C2['p'])
    with pytest.raises(ValueError, match="constant parameter"):
        nmpc.optimize([.1, 40., 0., 0.])
