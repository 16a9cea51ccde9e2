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
out = (C.c_longlong * 32)()
f(None, 1)
u = nmpc.optimize(x, cp=p); torch.cuda.synchronize()
f(out, 0)
it = int(nmpc._nlp_solution['iter_count'][0]) + 1
print('derivative evaluations', it, [round(v / it) for v in out[:8]], '(0: stage points / their loads, 4: iterate + scalings, 5: cost terms)')
print('riccati per call: backward loop, x0 part, forward sweep, recovery', [round(v / (it - 1)) for v in out[8:12]])
print('errors: slot loops, reductions + scaling, tolerance test + barrier update', [round(v / it) for v in out[12:15]])
print('step: slot loop, reductions (+ barrier logs)', [round(v / (it - 1)) for v in out[15:17]])
print('line search per iteration: form_trial, eval_values, acceptance tests', [round(v / (it - 1)) for v in out[17:20]])
print('update', round(out[20] / (it - 1)))
