#!/usr/bin/env python3
"""Generate tests/golden/nmpc_*.json with the ORACLE (oracle/nmpc.py) - inputs and expected outputs only.

The reference holds no numeric NMPC assertions and cannot run here, so these fixtures are produced by the
build's own restatement after it passed the scipy cross-check of tests/test_oracle_nmpc.py ("parity unpinned").
Two correct solvers stopped at IPOPT's default tolerance (1e-8) differ by up to ~1e-5 in weakly curved directions of
this problem (the reduced Hessian of C2 is nearly singular at the solution: the inertia correction stays active), so
every step is also stored at tol = 1e-9 ('tight').

    python tests/golden/make_nmpc_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.nmpc import DenseIpm, NmpcProblem, IpmOptions          # noqa: E402
from oracle import models                               # noqa: E402
from tests.problems import C2, c2_x0, oracle_problem   # noqa: E402


def dump(name, spec_note, x0, p, res_list):
    out = {'note': spec_note, 'x0': x0.tolist(), 'p': np.asarray(p).tolist(), 'steps': []}
    for r, rt in res_list:
        out['steps'].append({'x0': r['x0'].tolist(), 'v_opt': r['v'].tolist(), 'u0': r['u0'].tolist(),
                             'f': r['f'].tolist(), 'status': r['status'].tolist(), 'iters': r['iters'].tolist(),
                             'kkt': r['kkt'].tolist(),
                             'tight': {'tol': 1e-9, 'v_opt': rt['v'].tolist(), 'u0': rt['u0'].tolist(),
                                       'f': rt['f'].tolist(), 'kkt': rt['kkt'].tolist()}})
    with open(os.path.join(HERE, name), 'w') as f:
        json.dump(out, f)
    print(name, [s['status'] for s in out['steps']])


def closed_loop(pb, x0, p, n_steps, u_old=None):
    """Each step is solved twice from the same warm start: at IPOPT's default tol = 1e-8 (status / iteration parity,
    drives the closed loop) and at tol = 1e-9 ('tight': the KKT point to ~1e-7, for the north-star comparison)."""
    ipm, ipm_t = DenseIpm(pb), DenseIpm(pb, IpmOptions(tol=1e-9))
    res_list, w, x = [], None, x0
    for _ in range(n_steps):
        r = ipm.solve(x, p, w0=w, u_old=u_old)
        rt = ipm_t.solve(x, p, w0=w, u_old=u_old)
        for q in (r, rt):
            q['x0'], q['v'] = x, ipm.to_v(q)
        res_list.append((r, rt))
        w = r['w']
        if u_old is not None:
            u_old = r['U'][:, 0]
        x = pb.phi(x / pb.sx, r['U'][:, 0], p) * pb.sx
    return res_list


def main():
    pb = oracle_problem(C2)
    x0 = c2_x0(8)
    dump('nmpc_c2.json', 'C2 (tests/problems.py), 8 instances, 3 closed-loop steps, warm-started un-shifted',
         x0, C2['p'], closed_loop(pb, x0, C2['p'], 3))
    spec = dict(C2, x_scaling=[.1, 40., 2., 1.], u_scaling=[2., 2.])
    pb = oracle_problem(spec)
    dump('nmpc_c2_scaled.json', 'C2 with x_scaling=[.1,40,2,1], u_scaling=[2,2]', x0[:4], C2['p'],
         closed_loop(pb, x0[:4], C2['p'], 2))
    pb = NmpcProblem(models.get('pendulum4'), dt=.1, N=25, order=4,
                     stage_states=[([1, 2], [10., 5.], [0., 0.])], stage_inputs=[([0], [.1], None)],
                     input_change=([0], [1.]), x_lb=[-5, -10, -10, -10], x_ub=[5, 10, 10, 10],
                     x_guess=[2.5, 0., .1, 0.], u_guess=[0.])
    rng = np.random.default_rng(7)
    xp = np.array([2.5, 0., .1, 0.]) + .05 * rng.normal(size=(4, 4))
    dump('nmpc_pendulum.json', 'pendulum of reference tests/test_NMPC.py:12-67 (N=25, weights 10/5/0.1, box on x) plus an '
         'input-change penalty 1.0 on F', xp, [], closed_loop(pb, xp, np.zeros((4, 0)), 2, u_old=np.zeros((4, 1))))


if __name__ == '__main__':
    main()
