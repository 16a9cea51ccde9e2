"""GPU box: multiplier error of the DAE test problem against the oracle (debugging aid)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.test_dae_gpu as t
x0 = np.array([[2.5, 0., .1, 0.], [2., .2, -.1, .1]])
for tol in (1e-10, 1e-12):
    pb, ipm = t._pendulum_oracle(tol=tol)
    nmpc = t._pendulum(tol=tol)
    ref = ipm.solve(x0, [])
    nmpc.optimize(x0)
    v, vr = nmpc._nlp_solution['x'].cpu().numpy(), ipm.to_v(ref)
    lam = nmpc._nlp_solution['lam_g'].cpu().numpy()
    err = np.abs(lam - ref['lam']) / (1e-7 / 1e-5 + np.abs(ref['lam']))
    print('tol', tol, 'iters', nmpc.stats()['iter_count'], ref['iters'], 'v err', np.max(np.abs(v - vr) / np.maximum(1., np.abs(vr))),
          'lam rel err max', err.max(), 'median', np.median(err), 'kkt', nmpc.stats()['kkt_error'])
