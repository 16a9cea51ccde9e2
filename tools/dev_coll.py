"""Developer loop for the collocation policy without a GPU: compile a general problem of tests/problems.py with hiprtc
(HILO_JIT_COMPILE_ONLY=1, private cache directory) and print the solve kernel's register / scratch / LDS figures from the code
object's notes.

    python tools/dev_coll.py [C5D|C5DS|...] [extra hiprtc options]
"""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'


def main(argv):
    name = argv[0] if argv else 'C5D'
    cache = os.environ.setdefault('HILO_JIT_CACHE', '/tmp/jc_dev')
    os.makedirs(cache, exist_ok=True)
    os.environ['HILO_JIT_COMPILE_ONLY'] = '1'
    if len(argv) > 1:
        os.environ['HILO_JIT_EXTRA_OPTS'] = ' '.join(argv[1:])
    before = set(os.listdir(cache))
    from tests import problems
    t0 = time.time()
    try:
        problems.product_gen(getattr(problems, name))
    except Exception as e:                                   # compile-only mode ends setup() with HILO_COMPILED_ONLY
        print('setup():', str(e)[:600])
    print(f'compiled in {time.time() - t0:.1f} s')
    for f in sorted(set(os.listdir(cache)) - before):
        if not f.endswith('.hsaco'):
            continue
        path = os.path.join(cache, f)
        notes = subprocess.run([READELF, '--notes', path], capture_output=True, text=True).stdout
        for blk in notes.split('- .agpr_count')[1:]:
            get = lambda key: (re.search(rf'\.{key}:\s*(\S+)', blk) or [None, '?'])[1]
            print(f"{get('name'):32s} vgpr {get('vgpr_count'):>4s} agpr {blk.split()[0].strip(':') if False else get('agpr_count') if False else '':s}"
                  f"sgpr {get('sgpr_count'):>4s} scratch {get('private_segment_fixed_size'):>6s} B  lds {get('group_segment_fixed_size'):>7s} B  "
                  f"spill {get('vgpr_spill_count')}")
        print(path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main(sys.argv[1:])
