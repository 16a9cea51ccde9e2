"""CPU: the LMPC oracle (reference QP assembly + dense QP solver) cross-checked by scipy on the same QP; pins the
integer bookkeeping and the Q5 `kron(B, I)` quirk."""
import numpy as np
import pytest
from scipy.optimize import minimize

from oracle.lmpc import LmpcProblem, lmpc_optimize, solve_qp

from tests.problems import C1, LMPC_A as A, LMPC_B as B, LMPC_DT as DT      # (the benchmark's configuration 1: tests/test_LMPC.py:8-33)


def test_layout_and_kron_quirk():
    pb = LmpcProblem(**C1)
    assert pb.n_v == 32 and pb.Aeq.shape == (20, 32)
    assert pb.x_ind[0] == [0, 1] and pb.u_ind[0] == [22] and pb.u_ind[9] == [31]
    # Q5: kron(B, I_N) puts B[0] on rows 0..9 and B[1] on rows 10..19 of the input block
    np.testing.assert_array_equal(pb.Aeq[:10, 22:], B[0, 0] * np.eye(10))
    np.testing.assert_array_equal(pb.Aeq[10:, 22:], B[1, 0] * np.eye(10))
    ok = LmpcProblem(**C1, kron_bug=False)
    np.testing.assert_array_equal(ok.Aeq[0:2, 22], B[:, 0])
    np.testing.assert_array_equal(ok.Aeq[2:4, 23], B[:, 0])
    # H = blkdiag(I (x) Q, P, I (x) R) with P = 0 by default (mpc.py:2188-2193)
    assert np.all(np.diag(pb.H)[:20] == 1) and np.all(np.diag(pb.H)[20:22] == 0) and np.all(np.diag(pb.H)[22:] == 1)


@pytest.mark.parametrize('bug', [True, False])
def test_qp_vs_scipy(bug):
    from scipy.optimize import linprog
    pb = LmpcProblem(**C1, kron_bug=bug)
    rng = np.random.default_rng(3)
    n_feasible = 0
    for x0 in np.vstack([[1., 1.], rng.uniform(-4, 4, (5, 2))]):
        lb, ub = pb.bounds_for(x0)
        res = solve_qp(pb.H, pb.g, pb.Aeq, pb.beq, lb, ub)
        lp = linprog(np.zeros(pb.n_v), A_eq=pb.Aeq, b_eq=pb.beq, bounds=list(zip(lb, ub)))
        if lp.status == 2:
            # some measured states make the QP infeasible (box on x; more often with the scrambled block of mpc.py:2243)
            assert res['status'] == 3 and res['iters'] <= 20           # reported, not iterated to the cap
            continue
        n_feasible += 1
        assert res['status'] == 1
        x = res['x']
        assert np.abs(pb.Aeq @ x).max() < 1e-9 and np.all(x >= lb - 1e-9) and np.all(x <= ub + 1e-9)
        # stationarity in CasADi's sign convention: H x + g + A^T lam_a + lam_x = 0
        assert np.abs(pb.H @ x + pb.g + pb.Aeq.T @ res['y'] + res['z']).max() < 1e-8
        free = lb != ub
        full = lambda w: np.where(free, 0, lb) + np.concatenate([np.zeros(2), w]) if False else np.concatenate([lb[:2], w])  # noqa
        sol = minimize(lambda w: .5 * full(w) @ pb.H @ full(w), x[2:], jac=lambda w: (pb.H @ full(w))[2:], method='SLSQP',
                       bounds=list(zip(lb[2:], ub[2:])),
                       constraints=[{'type': 'eq', 'fun': lambda w: pb.Aeq @ full(w), 'jac': lambda w: pb.Aeq[:, 2:]}],
                       options={'ftol': 1e-14, 'maxiter': 300})
        np.testing.assert_allclose(sol.fun, res['f'], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(sol.x, x[2:], atol=2e-5)
    assert n_feasible >= 3


def test_closed_loop_double_integrator():
    """tests/test_LMPC.py:21-33 pattern with the corrected input block: the loop is stabilised."""
    pb = LmpcProblem(**C1, kron_bug=False)
    x = np.array([[1., 1.]])
    for _ in range(60):
        r = lmpc_optimize(pb, x)
        assert r['status'][0] == 1
        x = x @ A.T + r['u'] @ B.T
    assert np.abs(x).max() < 1e-3


def test_infeasible_states_are_reported_early():
    """OOQP's termination rule in the predictor-corrector iteration: a measured state from which the box cannot be kept makes the
    QP infeasible (scipy's phase 1 agrees); the iteration reports status 3 within a few steps and never does so on a feasible QP."""
    from scipy.optimize import linprog
    pb = LmpcProblem(**C1, kron_bug=False)
    rng = np.random.default_rng(8)
    xs = rng.uniform(-4, 4, (60, 2))
    res = lmpc_optimize(pb, xs)
    n_inf = 0
    for x0, st, it in zip(xs, res['status'], res['iters']):
        lb, ub = pb.bounds_for(x0)
        lp = linprog(np.zeros(pb.n_v), A_eq=pb.Aeq, b_eq=pb.beq, bounds=list(zip(lb, ub)))
        assert (st == 3) == (lp.status == 2) and (st == 1) == (lp.status == 0)
        if st == 3:
            n_inf += 1
            assert it <= 15
    assert n_inf >= 5
