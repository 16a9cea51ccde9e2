"""Developer aid: closed-loop timing of the GP-hybrid NMPC (C4) at its per-GPU batch size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.problems import C4, c2_x0, product_nmpc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nmpc = product_nmpc(C4)
x = torch.as_tensor(c2_x0(B), device='cuda'); p = torch.as_tensor(np.array(C4['p']), device='cuda')
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for s in range(12):
    ev[0].record()
    u = nmpc.optimize(x, cp=p)
    ev[1].record(); torch.cuda.synchronize()
    it = nmpc._nlp_solution['iter_count'].cpu().numpy(); st = nmpc._nlp_solution['status'].cpu().numpy()
    print(s, 'ms %.3f' % ev[0].elapsed_time(ev[1]), 'iters mean %.2f max %d' % (it.mean(), it.max()),
          'status', dict(zip(*np.unique(st, return_counts=True))))
    x = nmpc.plant_step(x, u, cp=p)
nmpc.phase_profile(True)
u = nmpc.optimize(x, cp=p); torch.cuda.synchronize()
pr = nmpc.phase_profile(True)
it = int(nmpc._nlp_solution['iter_count'][0])
print('iters', it, {k: (v if k.startswith('n_') else round(v / it)) for k, v in pr.items()})
