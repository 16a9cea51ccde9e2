"""Developer aid: iteration-count distribution of the C2 closed loop (the kernel runs as long as its slowest instance)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.problems import C2, c2_x0, product_nmpc
B = 1024
nmpc = product_nmpc(C2)
x = torch.as_tensor(c2_x0(B), device='cuda'); p = torch.as_tensor(np.array(C2['p']), device='cuda')
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for s in range(30):
    ev[0].record()
    u = nmpc.optimize(x, cp=p)
    ev[1].record(); torch.cuda.synchronize()
    it = nmpc._nlp_solution['iter_count'].cpu().numpy(); st = nmpc._nlp_solution['status'].cpu().numpy()
    print(s, 'ms %.3f' % ev[0].elapsed_time(ev[1]), 'iters mean %.2f p50 %d p90 %d p99 %d max %d' % (it.mean(), *np.percentile(it, [50, 90, 99]), it.max()),
          'status', dict(zip(*np.unique(st, return_counts=True))))
    x = nmpc.plant_step(x, u, cp=p)
