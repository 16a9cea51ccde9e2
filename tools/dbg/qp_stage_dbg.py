"""Debug aid: the stage QP kernel against the dense one after 1, 2, 3 .. iterations (max_iter), single instance."""
import os
import sys
import numpy as np
sys.path.insert(0, '.')
from tests.test_lmpc_gpu import product_lmpc                          # noqa: E402
from hilo_mpc_amd import LMPC, Model                                   # noqa: E402
from tests.test_oracle_lmpc import A, B, DT                            # noqa: E402


def make(mi, dense, N=int(os.environ.get('NH', 10))):
    if dense:
        os.environ['HILO_QP_DENSE'] = '1'
    else:
        os.environ.pop('HILO_QP_DENSE', None)
    m = Model('lti', A=A, B=B).setup(dt=DT)
    mpc = LMPC(m)
    mpc.Q, mpc.R, mpc.horizon = np.eye(2), 1, N
    mpc.set_box_constraints(x_lb=[-5, -5], x_ub=[5, 5], u_lb=[-1], u_ub=[1])
    mpc.setup(kron_variant='corrected', solver_options={'max_iter': mi})
    return mpc


x0 = np.array([[1., 1.], [.5, -.3]])
np.set_printoptions(precision=5, linewidth=200, suppress=True)
for mi in (1, 2, 3, 8, 100):
    a, b = make(mi, False), make(mi, True)
    assert a._qp_stages and not b._qp_stages
    a.optimize(x0)
    b.optimize(x0)
    sa, sb = a._nlp_solution, b._nlp_solution
    print('max_iter', mi, 'status', a.solver_status_code, b.solver_status_code, 'iters', sa['iter_count'].cpu().numpy(), sb['iter_count'].cpu().numpy())
    for key in ('x', 'lam_a', 'lam_x', 'f'):
        va, vb = sa[key].cpu().numpy(), sb[key].cpu().numpy()
        print('   ', key, 'diff', np.abs(va - vb).max())
        if mi == 1 and key != 'f':
            print('      stage', va[0])
            print('      dense', vb[0])
