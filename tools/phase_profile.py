"""Developer aid: per-phase shader-clock cycles per interior-point iteration of instance 0 (C2 steady state)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.problems import C2, c2_x0, product_nmpc
B = 1024
nmpc = product_nmpc(C2)
x = torch.as_tensor(c2_x0(B), device='cuda'); p = torch.as_tensor(np.array(C2['p']), device='cuda')
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    u = nmpc.optimize(x, cp=p); x = nmpc.plant_step(x, u, cp=p)
nmpc.phase_profile(True)
u = nmpc.optimize(x, cp=p); torch.cuda.synchronize()
pr = nmpc.phase_profile(True)
it = int(nmpc._nlp_solution['iter_count'][0])
print('iters', it, {k: (v if k.startswith('n_') else round(v / it)) for k, v in pr.items()},
      'sum', round(sum(v for k, v in pr.items() if not k.startswith('n_')) / it))
