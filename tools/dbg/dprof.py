"""Developer aid (build with HILO_EXTRA_FLAGS=-DHILO_OCP_DPROF): cycles of the sections of eval_derivs_sym, instance 0."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hilo_mpc_amd import _lib
from tests.problems import C2, c2_x0, product_nmpc
nmpc = product_nmpc(C2)
x = torch.as_tensor(c2_x0(1024), device='cuda'); p = torch.as_tensor(np.array(C2['p']), device='cuda')
for _ in range(3):
    u = nmpc.optimize(x, cp=p); x = nmpc.plant_step(x, u, cp=p)
torch.cuda.synchronize()
f = _lib.lib().hilo_debug_dprof
f.argtypes = [C.c_void_p, C.c_int]
out = (C.c_longlong * 16)()
f(None, 1)
u = nmpc.optimize(x, cp=p); torch.cuda.synchronize()
f(out, 0)
it = int(nmpc._nlp_solution['iter_count'][0]) + 1
print('derivative evaluations', it, [round(v / it) for v in out[:8]])
print('riccati per call: backward loop, x0 part, forward sweep, recovery', [round(v / (it - 1)) for v in out[8:12]])
