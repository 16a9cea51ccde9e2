"""Developer aid: C5-DAE at N = 50 with and without the shared collocation factors (NmpcUser::PREP): same numbers, time per solve.
    HILO_JIT_CACHE=/tmp/c1 python tools/dbg/c5dae_prep.py out1.npy
    HILO_JIT_CACHE=/tmp/c2 HILO_JIT_EXTRA_OPTS=-DHILO_USER_NO_PREP HILO_NO_PREP_WS=1 python tools/dbg/c5dae_prep.py out2.npy"""
import sys
import time
import numpy as np
import torch
sys.path.insert(0, '.')
from tests.problems import C5D, c5_x0, product_gen

B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nmpc = product_gen(C5D)
x0 = c5_x0(B)
u = nmpc.optimize(x0)
torch.cuda.synchronize()
t = time.time()
x1 = nmpc.plant_step(x0, u).cpu().numpy()
u = nmpc.optimize(x1)
torch.cuda.synchronize()
dt = time.time() - t
st = nmpc.stats()
print('B', B, 'second step', dt, 's', 'iters mean', st['iter_count'].mean(), 'status', np.unique(nmpc.solver_status_code, return_counts=True))
np.save(sys.argv[1], nmpc._nlp_solution['x'].cpu().numpy())
