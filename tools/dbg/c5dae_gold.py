"""Developer aid: product vs the N = 50 fixture of configuration 5 on the DAE at two tolerances."""
import json, sys
import numpy as np
sys.path.insert(0, '.')
from tests.problems import C5D, product_gen
g = json.load(open('tests/golden/nmpc_c5dae.json'))
x0, vr, lr, fr = np.array(g['x0']), np.array(g['v']), np.array(g['lam_g']), np.array(g['f'])
for tol in (1e-9, 1e-10, 1e-11):
    nmpc = product_gen(C5D, **{'ipopt.tol': tol})
    for start in ('oracle', 'guess'):
        nmpc.reset_solution()
        u = nmpc.optimize(x0, v0=vr if start == 'oracle' else None)
        v = nmpc._nlp_solution['x'].cpu().numpy()
        lam = nmpc._nlp_solution['lam_g'].cpu().numpy()
        print(tol, start, 'status', nmpc.solver_status_code, 'iters', nmpc.stats()['iter_count'], 'kkt', nmpc.stats()['kkt_error'],
              'v', (np.abs(v - vr) / np.maximum(1., np.abs(vr))).max(1), 'lam', (np.abs(lam - lr) / np.maximum(1., np.abs(lr))).max(1),
              'f', nmpc._nlp_solution['f'].cpu().numpy() - fr)
