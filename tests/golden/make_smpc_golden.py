"""Fixture of the stochastic-NMPC parity tests: outputs of the ORACLE (oracle/smpc.py: sympy restatement of the reference's
surrogate + the dense interior point) for the cases of tests/problems.py::SMPC_CASES, so that the GPU tests do not spend their
time in sympy.  (The reference's own SMPC tests, tests/test_SMPC.py, hold no numbers; CasADi is not installable here.)

    python tests/golden/make_smpc_golden.py        # writes tests/golden/smpc.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.nmpc import IpmOptions                                                 # noqa: E402
from oracle.nmpc_gen import GenIpm                                                 # noqa: E402
from oracle.smpc import smpc_surrogate                                             # noqa: E402
from tests.problems import SMPC_CASES, smpc_models, smpc_oracle_post, smpc_oracle_problem   # noqa: E402


def main():
    out = {}
    post = smpc_oracle_post()
    for name, c in SMPC_CASES.items():
        rng = np.random.default_rng(len(name))
        _, om = smpc_models(name)
        n, nu = om.nx, om.nu
        K = np.asarray(c['K'], dtype=float)
        # ---- the surrogate map at random points (gain as parameters) ----
        sur, _, _ = smpc_surrogate(om, [post], [c['features']], c['Bw'], None)
        pts = []
        for _ in range(8):
            mean = np.asarray(c['x0']) * (1 + .3 * rng.uniform(-1, 1, n)) + rng.uniform(-.5, .5, n)
            if name != 'pend':
                mean[0] = rng.uniform(-.5, 1.5)
            A = rng.uniform(-1, 1, (n, n))
            cov = .1 * A @ A.T
            u = rng.uniform(-1, 1, nu)
            p = (K * (1 + .5 * rng.uniform(-1, 1, K.shape))).T.reshape(-1)
            xa = np.concatenate([mean, cov.T.reshape(-1)])
            pts.append(dict(xa=xa.tolist(), u=u.tolist(), p=p.tolist(), f=sur.f(xa, u, p, 1.)[0].tolist()))
        if not c.get('solve', True):
            out[name] = dict(points=pts)
            continue
        # ---- solves: the nominal start and perturbed ones, default tolerance and a tight one ----
        B = 4
        x0 = np.asarray(c['x0'], dtype=float) * (1 + .03 * rng.uniform(-1, 1, (B, n)))
        x0[0] = c['x0']
        cov0 = np.asarray(c['cov0'], dtype=float)
        xa0 = np.concatenate([x0, np.tile(cov0.T.reshape(-1), (B, 1))], axis=1)
        p = np.tile(K.T.reshape(-1), (B, 1))
        pb = smpc_oracle_problem(name)
        sol = {}
        for tag, tol in (('default', None), ('tight', 1e-10)):
            o = IpmOptions()
            if tol:
                o.tol = tol
            ipm = GenIpm(pb, o)
            r = ipm.solve(xa0, p)
            sol[tag] = dict(status=r['status'].tolist(), f=r['f'].tolist(), v=ipm.to_v(r).tolist(), u0=r['u0'].tolist(),
                            iters=np.asarray(r['iters']).tolist() if 'iters' in r else None)
            print(name, tag, r['status'], r['f'])
        out[name] = dict(points=pts, x0=x0.tolist(), n_v=pb.n_v, n_g=pb.n_g, solves=sol)
    with open(os.path.join(ROOT, 'tests', 'golden', 'smpc.json'), 'w') as f:
        json.dump(out, f)


if __name__ == '__main__':
    main()
