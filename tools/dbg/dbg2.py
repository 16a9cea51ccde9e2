import sys; sys.path.insert(0, '.')
import numpy as np
from tests.problems import cstr_nmpc, cstr_plant
nmpc = cstr_nmpc()
x = np.array([[1., 0., 400.]])
for k in range(250):
    u = nmpc.optimize(x); x = cstr_plant(x, u)
print('after 250: Q %.6f T %.8f iters' % (u[0, 0], x[0, 2]), nmpc._nlp_solution['iter_count'].cpu().numpy(), nmpc._nlp_solution['kkt_error'].cpu().numpy())
v_warm = nmpc._nlp_solution['x'].clone()
nmpc.phase_profile(True)
u = nmpc.optimize(x, v0=v_warm)
print('profile', nmpc.phase_profile(True), 'iters', nmpc._nlp_solution['iter_count'].cpu().numpy())
for tol in (1e-8, 1e-9, 1e-10, 1e-12):
    n2 = cstr_nmpc(tol=tol)
    u2 = n2.optimize(x, v0=v_warm)
    print('tol %g: Q %.6f iters %s kkt %s st %s' % (tol, u2[0, 0], n2._nlp_solution['iter_count'].cpu().numpy(), n2._nlp_solution['kkt_error'].cpu().numpy(), n2.solver_status_code))
for mi in range(1, 10):
    n2 = cstr_nmpc(max_iter=mi)
    u2 = n2.optimize(x, v0=v_warm)
    print('max_iter %d: Q %.6f kkt %s st %s' % (mi, u2[0, 0], n2._nlp_solution['kkt_error'].cpu().numpy(), n2.solver_status_code))
