import sys; sys.path.insert(0, '.')
import numpy as np, torch
from tests import problems as P
gp = P.product_gp()
m = 1 << 18
rng = np.random.default_rng(3)
Xq = torch.as_tensor(np.stack([rng.uniform(0, 40, m), rng.uniform(0, 4, m)]), device='cuda')
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print('mean only  ms', timeit(lambda: gp.predict(Xq, return_var=False)))
print('mean + var ms', timeit(lambda: gp.predict(Xq)))
