"""Developer aid: per-phase shader-clock cycles per interior-point iteration of instance 0, closed loop steady state.
   python tools/phase_profile.py [warm-up steps] [C2 | C4] [batch]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.problems import C2, C4, c2_x0, product_nmpc
nwarm = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = sys.argv[2] if len(sys.argv) > 2 else 'C2'
B = int(sys.argv[3]) if len(sys.argv) > 3 else (1024 if cfg == 'C2' else 2048)
spec = C2 if cfg == 'C2' else C4
nmpc = product_nmpc(spec)
x = torch.as_tensor(c2_x0(B), device='cuda'); p = torch.as_tensor(np.array(spec['p']), device='cuda')
for _ in range(nwarm):
    u = nmpc.optimize(x, cp=p); x = nmpc.plant_step(x, u, cp=p)
torch.cuda.synchronize()
nmpc.phase_profile(True)
t0 = time.time()
u = nmpc.optimize(x, cp=p); torch.cuda.synchronize()
dt = time.time() - t0
pr = nmpc.phase_profile(True)
it = int(nmpc._nlp_solution['iter_count'][0])
its = nmpc._nlp_solution['iter_count'].double()
print(cfg, 'B', B, 'launch ms', round(dt * 1e3, 3), 'iters inst0', it, 'mean', round(float(its.mean()), 2), 'max', float(its.max()),
      {k: (v if k.startswith('n_') else round(v / max(it, 1))) for k, v in pr.items()},
      'sum', round(sum(v for k, v in pr.items() if not k.startswith('n_')) / max(it, 1)))
