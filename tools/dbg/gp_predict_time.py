"""Developer aid (GPU): host-side launch time vs completed time of `GaussianProcess.predict` at 2^18 queries, through the class and
through the raw C ABI."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import problems as P
gp = P.product_gp()
m = 1 << 18
rng = np.random.default_rng(3)
Xq = torch.as_tensor(np.stack([rng.uniform(0, 40, m), rng.uniform(0, 4, m)]), device='cuda')
for _ in range(5): gp.predict(Xq)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20): gp.predict(Xq)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('launch loop', (t1 - t0) / 20 * 1e3, 'ms/step; incl sync', (t2 - t0) / 20 * 1e3)
from hilo_mpc_amd import _lib
mean = torch.empty(1, m, dtype=torch.float64, device='cuda'); var = torch.empty_like(mean)
from hilo_mpc_amd.gp import ptr, stream_ptr
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    _lib.check(_lib.lib().hilo_gp_predict(gp._handle, m, ptr(Xq), 0, ptr(mean), ptr(var), stream_ptr(torch.device('cuda:0'))))
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('raw ABI loop', (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3)
