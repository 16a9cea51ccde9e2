#!/usr/bin/env python3
"""Generate tests/golden/nmpc_c4.json with the ORACLE: GP-hybrid NMPC (tests/problems.py C4) - inputs and expected
outputs only ("parity unpinned": the reference's hybrid example, nmpc_hybrid_bio.ipynb, asserts no numbers and CasADi
cannot be installed).  The training set of the GP is part of the fixture so that the test does not depend on the
random generator that produced it.  Only instances the oracle solves to status 1 in every step are kept.

    python tests/golden/make_hybrid_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden.make_nmpc_golden import closed_loop          # noqa: E402
from tests.problems import C4, C4_GP, c2_x0, c4_training_data, oracle_c4   # noqa: E402


def main():
    pb, post = oracle_c4()
    x0 = c2_x0(12)
    res = closed_loop(pb, x0, C4['p'], 2)
    keep = np.ones(x0.shape[0], dtype=bool)
    for r, rt in res:
        keep &= (r['status'] == 1) & (rt['status'] == 1)
    idx = np.nonzero(keep)[0][:6]
    X, y = c4_training_data()
    out = {'note': 'C4 (tests/problems.py): chemostat4 with mu <- GP mean over (S, I); 2 closed-loop steps, warm-started',
           'gp': dict(C4_GP, X=X.tolist(), y=y.tolist(), alpha=post.alpha.tolist(), lml=float(post.lml)),
           'x0': x0[idx].tolist(), 'p': list(C4['p']), 'steps': []}
    for r, rt in res:
        out['steps'].append({'x0': r['x0'][idx].tolist(), 'v_opt': r['v'][idx].tolist(), 'u0': r['u0'][idx].tolist(),
                             'f': r['f'][idx].tolist(), 'status': r['status'][idx].tolist(),
                             'iters': r['iters'][idx].tolist(), 'kkt': r['kkt'][idx].tolist(),
                             'tight': {'tol': 1e-9, 'v_opt': rt['v'][idx].tolist(), 'u0': rt['u0'][idx].tolist(),
                                       'f': rt['f'][idx].tolist(), 'kkt': rt['kkt'][idx].tolist()}})
    with open(os.path.join(HERE, 'nmpc_c4.json'), 'w') as f:
        json.dump(out, f)
    print('kept', idx.tolist(), [s['status'] for s in out['steps']], [s['iters'] for s in out['steps']])


if __name__ == '__main__':
    main()
