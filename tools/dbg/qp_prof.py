"""GPU box, developer build libhilo_hip_qpprof.so (-DHILO_QP_PROF): clock ticks per section of the LMPC QP iteration."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ['HILO_LIB_PATH'] = os.path.join(ROOT, 'hilo_mpc_amd', 'libhilo_hip_qpprof.so')
from hilo_mpc_amd import _lib  # noqa: E402
from tests.test_lmpc_gpu import product_lmpc  # noqa: E402

NAMES = ['(after update)', 'residuals', 'M rows + chol', 'columns L^-1, X', 'Schur rows', 'chol S', 'inverse factor S', 'two solves + step']
lib = _lib.lib()
for B in (1, 1024):
    mpc = product_lmpc('corrected')
    x = torch.as_tensor(np.tile([[1., 1.]], (B, 1)), device='cuda')
    mpc.optimize(x)
    out = (ctypes.c_longlong * 16)()
    lib.hilo_qp_debug_prof(out)
    mpc.optimize(x)
    lib.hilo_qp_debug_prof(out)
    it = int(mpc._nlp_solution['iter_count'][0])
    v = np.array(out[:8], dtype=float)
    print(f"B={B} iterations {it}; ticks per iteration (instance 0):")
    for n, c in zip(NAMES, v):
        print(f"  {n:24s} {c / max(it, 1):10.0f}")
    print(f"  {'sum':24s} {v.sum() / max(it, 1):10.0f}")
