"""Developer aid (CPU): the oracle's interior point, verbose, on a straggler of configuration 4's closed loop dumped by
tools/dbg/c4_tail_dump.py (gpurun_out/c4_tail.npz): does the reference algorithm need that many iterations from this warm start?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.nmpc import DenseIpm
from tests.problems import C4, oracle_c4
d = np.load('gpurun_out/c4_tail.npz')
step = int(sys.argv[1]) if len(sys.argv) > 1 else 5
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pb, _ = oracle_c4()
ipm = DenseIpm(pb)
x0, v0 = d[f'x0_{step}'][which:which + 1], d[f'v0_{step}'][which:which + 1]
print('device iterations', d[f'it_{step}'][which])
res = ipm.solve(x0, C4['p'], w0=ipm.w_from_v(v0), verbose=True)
print('oracle iterations', res['iters'], 'status', res['status'], 'f', res['f'], 'device f', d[f'f_{step}'][which])
