"""Golden vectors of BASELINE configuration 5 as it is written (tests/problems.py::C5D: path following on the robot's DAE with
the soft limit on the algebraic state, collocation Radau 3, N = 50) from the dense oracle (oracle/nmpc_coll_gen.py) at tol 1e-10:
B = 3 instances - the dense solve of the 1902-variable / 1750-row NLP takes minutes, too long for the GPU suite.

    python tests/golden/make_c5dae_golden.py        ->  tests/golden/nmpc_c5dae.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.nmpc import IpmOptions                                  # noqa: E402
from oracle.nmpc_coll_gen import GenCollIpm                         # noqa: E402
from tests.problems import C5D, c5_x0, oracle_coll_gen              # noqa: E402


def main():
    x0 = c5_x0(3)
    pb = oracle_coll_gen(C5D)
    ipm = GenCollIpm(pb, IpmOptions(tol=1e-10))
    res = ipm.solve(x0, [], verbose=True)
    out = dict(x0=x0.tolist(), n_v=pb.n_v, n_g=pb.n_g, status=res['status'].tolist(), iters=res['iters'].tolist(),
               f=res['f'].tolist(), kkt=res['kkt'].tolist(), u0=res['u0'].tolist(), v=ipm.to_v(res).tolist(),
               lam_g=ipm.lam_g(res).tolist())
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'nmpc_c5dae.json'), 'w') as f:
        json.dump(out, f)
    print('status', res['status'], 'iters', res['iters'], 'kkt', res['kkt'])


if __name__ == '__main__':
    main()
