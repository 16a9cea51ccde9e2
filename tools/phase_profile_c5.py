"""Developer aid: per-phase shader-clock cycles per interior-point iteration of instance 0 for the C5 workload (path following,
N = 50, general run-time compiled policy, iterate in the global workspace).   python tools/phase_profile_c5.py [batch]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.problems import C5, C5D, c5_x0, product_gen
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nmpc = product_gen(C5D if os.environ.get("C5DAE") else C5, **({"ipopt.max_iter": 12} if os.environ.get("HILO_DBG_ONE") else {}))
x = torch.as_tensor(c5_x0(B), device='cuda')
for _ in range(2):
    u = nmpc.optimize(x)
torch.cuda.synchronize()
nmpc.phase_profile(True)
t0 = time.time()
u = nmpc.optimize(x); torch.cuda.synchronize()
dt = time.time() - t0
pr = nmpc.phase_profile(True)
it = int(nmpc._nlp_solution['iter_count'][0])
its = nmpc._nlp_solution['iter_count'].double()
print('B', B, 'launch ms', round(dt * 1e3, 2), 'iters inst0', it, 'mean', float(its.mean()), 'max', float(its.max()),
      {k: (v if k.startswith('n_') else round(v / max(it, 1))) for k, v in pr.items()},
      'sum/iter', round(sum(v for k, v in pr.items() if not k.startswith('n_')) / max(it, 1)))
