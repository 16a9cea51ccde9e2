"""Fixture of the reference's only published NMPC result (docs/docsource/examples/CSTR_Example.ipynb, cells 4/6/14/16): the
oracle (oracle/nmpc_coll.py: the reference's simultaneous collocation NLP + the interior-point method of oracle/nmpc.py) runs
the notebook's 1000-step closed loop from [C_A, C_B, T] = [1, 0, 400]; the notebook prints

    True: Q: 59882.1817 C_A: 0.4912 C_B: 0.5088 T: 438.4732

Plant: the same ODE with 50 RK4 sub-steps per sampling interval (the reference calls CVODES; the loop converges to a steady
state that every consistent integrator shares).  Writes tests/golden/nmpc_cstr.json: the state / input every 50 steps, the
warm start at step 990 (so that the CPU suite can re-run the last ten steps in seconds), and the final line.

    python tests/golden/make_cstr_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.problems import CSTR, cstr_oracle, cstr_plant  # noqa: E402


def main():
    pb, ipm = cstr_oracle()
    x = np.array([CSTR['x0']], dtype=float)
    w, snaps, restart = None, [], None
    for k in range(1000):
        if k == 990:
            restart = {'step': k, 'x': x[0].tolist(), 'w': w[0].tolist()}
        res = ipm.solve(x, [], w0=w)
        assert res['status'][0] == 1, (k, res['status'])
        w, u = res['w'], res['u0']
        x = cstr_plant(x, u)
        if k % 50 == 49:
            snaps.append({'step': k + 1, 'x': x[0].tolist(), 'u': float(u[0, 0]), 'iters': int(res['iters'][0])})
    line = f"Q: {u[0, 0]:.4f} C_A: {x[0, 0]:.4f} C_B: {x[0, 1]:.4f} T: {x[0, 2]:.4f}"
    out = {'source': 'docs/docsource/examples/CSTR_Example.ipynb cell 16', 'printed_by_the_reference': 'Q: 59882.1817 C_A: 0.4912 '
           'C_B: 0.5088 T: 438.4732', 'printed_by_the_oracle': line, 'snapshots': snaps, 'restart': restart,
           'final': {'x': x[0].tolist(), 'u': float(u[0, 0])}}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'nmpc_cstr.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print(line)


if __name__ == '__main__':
    main()
