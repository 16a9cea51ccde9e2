"""Developer aid (GPU): milliseconds per `hilo_gp_refit` (kernel matrix + blocked Cholesky + alpha + LML) at n = 200."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import problems as P
gp = P.product_gp()
for _ in range(3): gp._device_refit()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): gp._device_refit()
torch.cuda.synchronize(); print('refit ms', (time.perf_counter() - t0) / 50 * 1e3, 'lml', gp.log_marginal_likelihood())
