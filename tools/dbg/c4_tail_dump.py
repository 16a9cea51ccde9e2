"""Developer aid (GPU): the stragglers of configuration 4's closed loop - start state, warm start and result of the slowest instances
of a step, written to gpurun_out/c4_tail.npz for a verbose run of the oracle's interior point on the CPU (tools/dbg/c4_tail_cpu.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.problems import C4, c2_x0, product_nmpc
B = 2048
nmpc = product_nmpc(C4)
x = torch.as_tensor(c2_x0(B), device='cuda'); p = torch.as_tensor(np.array(C4['p']), device='cuda')
out = {}
for s in range(8):
    v_prev = None if nmpc._nlp_solution is None else nmpc._nlp_solution['x'].clone()
    x_in = x.clone()
    u = nmpc.optimize(x, cp=p)
    it = nmpc._nlp_solution['iter_count'].cpu().numpy()
    print(s, 'iters mean %.2f max %d' % (it.mean(), it.max()), np.argsort(it)[-3:], np.sort(it)[-3:])
    if s >= 3 and v_prev is not None:
        idx = np.argsort(it)[-3:]
        out[f'x0_{s}'] = x_in.cpu().numpy()[idx]; out[f'v0_{s}'] = v_prev.cpu().numpy()[idx]
        out[f'it_{s}'] = it[idx]; out[f'v_{s}'] = nmpc._nlp_solution['x'].cpu().numpy()[idx]
        out[f'f_{s}'] = nmpc._nlp_solution['f'].cpu().numpy()[idx]
        ref = np.argsort(it)[:3]
        out[f'x0r_{s}'] = x_in.cpu().numpy()[ref]; out[f'v0r_{s}'] = v_prev.cpu().numpy()[ref]; out[f'itr_{s}'] = it[ref]
    x = nmpc.plant_step(x, u, cp=p)
os.makedirs('gpurun_out', exist_ok=True)
np.savez('gpurun_out/c4_tail.npz', **out)
