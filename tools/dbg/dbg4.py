import sys; sys.path.insert(0, '.')
import numpy as np, torch
from tests import problems as P
from tests.test_coll_gpu import _product
from tests.test_mhe_gpu import product_mhe
def frac(st): return dict(zip(*[a.tolist() for a in np.unique(st, return_counts=True)]))
# collocation full batch closed loop (continuous objective, JIT)
nm = _product(dict(P.C2)); x = torch.as_tensor(P.c2_x0(1024), device='cuda'); p = torch.as_tensor(np.array(P.C2['p']), device='cuda')
for k in range(4):
    u = nm.optimize(x, cp=p); print('coll step', k, frac(nm.solver_status_code), float(nm._nlp_solution['iter_count'].double().mean())); x = nm.plant_step(x, u, cp=p)
# C4 B=256 5 steps
nm = P.product_nmpc(P.C4); x = torch.as_tensor(P.c2_x0(256), device='cuda')
for k in range(5):
    u = nm.optimize(x, cp=p); print('C4 step', k, frac(nm.solver_status_code), float(nm._nlp_solution['iter_count'].double().mean())); x = nm.plant_step(x, u, cp=p)
# plain C3 B=4096
for spec, name in ((P.C3, 'C3 plain'), (P.C3B, 'C3B')):
    xa, u, y, _ = P.c3_data(4096, seed=11); mhe = product_mhe(spec)
    for k in range(spec['N']): mhe.add_measurements(torch.as_tensor(y[:, k], device='cuda'), torch.as_tensor(u[:, k], device='cuda'))
    mhe.estimate(x_arrival=torch.as_tensor(xa, device='cuda')); print(name, frac(mhe.solver_status_code))
# C5 from a zero-velocity guess, B=1024
spec = dict(P.C5, x_guess=[0., 0., 0., 0., 0., 0.])
nm = P.product_gen(spec); x = torch.as_tensor(P.c5_x0(1024), device='cuda')
for k in range(3):
    u = nm.optimize(x); print('C5 cold step', k, frac(nm.solver_status_code), float(nm._nlp_solution['iter_count'].double().mean())); x = nm.plant_step(x, u)
