"""Writes tests/golden/fullbatch_{c2,c3,c5}.json: the ORACLE's solutions (oracle/nmpc.py, oracle/mhe.py, oracle/nmpc_gen.py) of the
32-instance subsets of the BASELINE batches that tests/test_fullbatch_gpu.py compares the HIP path's full-batch launches with.  Run
in the build container (CPU only, a few minutes):   python tests/golden/make_fullbatch_golden.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ['HILO_RECOMPUTE_GOLDEN'] = '1'

from tests.util import golden_dump                                   # noqa: E402


def main():
    # (the functions live in the test module; importing it needs no GPU)
    import importlib.util
    spec = importlib.util.spec_from_file_location('fullbatch', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                              'test_fullbatch_gpu.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name, fn in (('fullbatch_c2', mod.oracle_c2), ('fullbatch_c3', mod.oracle_c3), ('fullbatch_c5', mod.oracle_c5)):
        data = fn()
        assert (data['status'] == 1).all(), (name, data['status'])
        print(name, golden_dump(name, data), 'iterations', float(data['iters'].mean()))


if __name__ == '__main__':
    main()
