"""Oracle-side fixtures of tests/test_mhe_gen_gpu.py for the cases whose oracle problem (the discretised model written out
symbolically with an estimated parameter / constraint rows) takes about a minute of sympy each: status, solution vector, objective,
multipliers, estimates and the index maps of oracle/mhe_gen.py on the seeded data of tests/problems.py::c3_data.

    python tests/golden/make_mhe_gen_fixtures.py          (CPU only; about five minutes)

The test calls the same functions when a fixture is missing (or with HILO_RECOMPUTE_GOLDEN=1)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if __name__ == '__main__':
    from tests.test_mhe_gen_gpu import hard_con_case, param_est_case
    from tests.util import golden_dump
    for noise in (False, True):
        print(golden_dump(f'mhe_gen_param_est_discrete_{int(noise)}', param_est_case('discrete', noise)))
        print(golden_dump(f'mhe_gen_hard_con_discrete_{int(noise)}', hard_con_case('discrete', noise)))
