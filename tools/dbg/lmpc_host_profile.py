"""Host time of `LMPC.optimize` at the configured batch (1024 QPs): wall time per call with and without the device, cProfile."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.problems import product_lmpc      # noqa: E402

dev = torch.device('cuda:0')
mpc = product_lmpc('corrected')
x = torch.as_tensor(np.random.default_rng(1).uniform(-1.5, 1.5, (1024, 2)), device=dev)
for _ in range(50):
    mpc.optimize(x)
torch.cuda.synchronize()
n = 3000
t = time.perf_counter()
for _ in range(n):
    mpc.optimize(x)
th = time.perf_counter() - t
torch.cuda.synchronize()
ta = time.perf_counter() - t
print(f'host {th / n * 1e6:.1f} us/call, with the device {ta / n * 1e6:.1f} us/call')
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    mpc.optimize(x)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
