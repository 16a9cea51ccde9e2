"""Developer aid: cycles of the Riccati sub-phases per CALL (a debug library built with -DHILO_OCP_DPROF and a call counter in
g_dprof[12]; experimental variants of the stage loop whose results may be garbage - only the timing is read)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hilo_mpc_amd import _lib
from tests.problems import C2, c2_x0, product_nmpc
nmpc = product_nmpc(C2)
x = torch.as_tensor(c2_x0(1024), device='cuda'); p = torch.as_tensor(np.array(C2['p']), device='cuda')
f = _lib.lib().hilo_debug_dprof
f.argtypes = [C.c_void_p, C.c_int]
out = (C.c_longlong * 16)()
u = nmpc.optimize(x, cp=p); torch.cuda.synchronize()
f(None, 1)
u = nmpc.optimize(x, cp=p); torch.cuda.synchronize()
f(out, 0)
n = max(1, out[12])
print(os.environ.get('HILO_LIB_PATH', 'product').split('_')[-1], 'riccati calls', out[12], 'per call: backward, x0, forward, recovery',
      [round(v / n) for v in out[8:12]], 'iters', int(nmpc._nlp_solution['iter_count'][0]))
