"""GPU box: launch time of the LMPC QP solve, register-resident kernel against the LDS-column kernel, for several horizons."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_lmpc_gpu import product_lmpc  # noqa: E402


def run(N, B, lo):
    rng = np.random.default_rng(5)
    x = torch.as_tensor(rng.uniform(-lo, lo, (B, 2)), device='cuda')
    mpc = product_lmpc('corrected', N=N)
    for _ in range(3):
        mpc.optimize(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        mpc.optimize(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    it = mpc._nlp_solution['iter_count'].double()
    st = mpc.solver_status_code
    return dt * 1e3, float(it.mean()), float(it.max()), float((st == 1).mean())


for N in (10, 14, 20):
    for B in (1, 1024):
        for col in ('0', '1'):
            if col == '1':
                os.environ['HILO_QP_LDS_COLUMNS'] = '1'
            else:
                os.environ.pop('HILO_QP_LDS_COLUMNS', None)
            ms, im, ix, ok = run(N, B, 1.5)
            print(f"N={N} B={B} lds_columns={col}: {ms:.3f} ms  iters mean {im:.1f} max {ix:.0f} solved {ok:.2f}  "
                  f"us/iter(max) {ms * 1e3 / ix:.1f}", flush=True)
