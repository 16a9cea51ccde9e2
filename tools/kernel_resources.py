#!/usr/bin/env python3
"""Register / scratch / LDS figures of every kernel in libhilo_hip.so (or the library given), from the code objects' metadata.
    python tools/kernel_resources.py [lib.so] [substring ...]
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith('.so') else os.path.join(HERE, '..', 'hilo_mpc_amd', 'libhilo_hip.so')
keys = [a for a in sys.argv[1:] if not a.endswith('.so')]
BIN = '/opt/rocm/lib/llvm/bin'
with tempfile.TemporaryDirectory() as td:
    data = open(lib, 'rb').read()
    # the bundles: concatenated "__CLANG_OFFLOAD_BUNDLE__" blobs inside .hip_fatbin; unbundle each by offset
    offs = [m.start() for m in re.finditer(b'__CLANG_OFFLOAD_BUNDLE__', data)]
    rows = []
    for n, o in enumerate(offs):
        end = offs[n + 1] if n + 1 < len(offs) else len(data)
        blob = os.path.join(td, f'b{n}.bin')
        open(blob, 'wb').write(data[o:end])
        co = os.path.join(td, f'b{n}.co')
        r = subprocess.run([f'{BIN}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={blob}', f'--output={co}',
                            '--targets=hipv4-amdgcn-amd-amdhsa--gfx950'], capture_output=True, text=True)
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        notes = subprocess.run([f'{BIN}/llvm-readelf', '--notes', co], capture_output=True, text=True).stdout
        for blk in notes.split('  - .agpr_count:')[1:]:
            get = lambda k: (re.search(rf'\.{k}:\s+(\S+)', blk) or [None, '?'])[1]   # noqa: E731
            name = get('name')
            dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
            rows.append((dem, get('vgpr_count'), re.match(r'\s*(\d+)', blk).group(1), get('private_segment_fixed_size'), get('group_segment_fixed_size')))
    for dem, v, a, sc, lds in sorted(rows):
        if not keys or any(k in dem for k in keys):
            print(f"vgpr {v:>4s} agpr {a:>4s} scratch {sc:>6s} B  lds {lds:>7s} B  {dem[:150]}")
