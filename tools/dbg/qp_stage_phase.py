"""Debug aid: cycle stamps of one iteration of the stage QP kernel (developer build -DHILO_QPO_PROF, HILO_LIB_PATH)."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from tests.test_lmpc_gpu import product_lmpc                          # noqa: E402
from hilo_mpc_amd import _lib                                         # noqa: E402

mpc = product_lmpc('corrected')
for _ in range(3):
    mpc.optimize(np.array([[1., 1.]]))
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
_lib.lib().hilo_debug_qpo_prof(out)
s = [int(v) for v in out]
print('iters', mpc._nlp_solution['iter_count'].cpu().numpy())
print('last full iteration, cycles: residuals+merit', s[1] - s[0], '| Sigma + factor sweep', s[2] - s[1], '| to pass 0', s[3] - s[2],
      '| pass0 sweeps', s[4] - s[3], 'dy+steps', s[5] - s[4], 'step reductions', s[6] - s[5],
      '| pass1 r1', s[7] - s[6], 'sweeps', s[8] - s[7], 'dy+steps', s[9] - s[8], 'reductions', s[10] - s[9], '| total', s[10] - s[0])
