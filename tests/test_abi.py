"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol the header declares, and the
host-side tables agree with the device zoo.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'hilo_hip.h')


def _declared():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(hilo_[a-z0-9_]+)\s*\(', src)))


def test_library_loads_and_exports_header_symbols():
    from hilo_mpc_amd import _lib
    lib = _lib.lib()
    assert lib.hilo_abi_version() == 1
    names = _declared()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/hilo_hip.h but not exported: {missing}"


def test_model_table_matches_device_zoo():
    from hilo_mpc_amd import _lib, Model
    from hilo_mpc_amd.model import ZOO
    lib = _lib.lib()
    for name in ZOO:
        if name in ('lti', 'chemostat4_gp'):          # caller-defined dims / built by Model.substitute_from
            continue
        m = Model(name)
        d = [C.c_int() for _ in range(5)]
        _lib.check(lib.hilo_model_dims(m.model_id, *[C.byref(v) for v in d]))
        assert [v.value for v in d] == [m.n_x, m.n_u, m.n_p, m.n_y, int(m._native_discrete)], name


def test_oracle_zoo_matches_product_zoo():
    from hilo_mpc_amd import Model
    from oracle import models
    for name in models.ZOO:
        om = models.get(name)
        if om.model_id < 0:          # oracle-only models (DAE variants): the product builds them from expressions
            continue
        pm = Model(name)
        assert (om.nx, om.nu, om.np_, om.ny) == (pm.n_x, pm.n_u, pm.n_p, pm.n_y), name
        assert om.model_id == pm.model_id or name == 'linear2'


def test_errors_are_reported_not_swallowed():
    from hilo_mpc_amd import _lib
    lib = _lib.lib()
    rc = lib.hilo_model_dims(999, None, None, None, None, None)
    assert rc == -1
    assert b'unknown model id' in lib.hilo_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)


def test_no_cpu_fallback_when_library_missing(monkeypatch):
    from hilo_mpc_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libhilo_hip.so')
    with pytest.raises(_lib.HiloError):
        _lib.lib()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'hilo_mpc_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', txt, flags=re.M), f


def test_descriptor_layouts_match_the_header(tmp_path):
    """The ctypes mirrors in hilo_mpc_amd/_lib.py against include/hilo_hip.h compiled by gcc: same field names, order,
    offsets and sizes for every descriptor struct (a drifted field would silently shift everything behind it)."""
    import ctypes as C
    import re
    import shutil
    import subprocess
    from hilo_mpc_amd import _lib
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    header = open(os.path.join(ROOT, 'include', 'hilo_hip.h')).read()
    pairs = [('hilo_kf_desc', _lib.KfDesc), ('hilo_nmpc_desc', _lib.NmpcDesc), ('hilo_mhe_desc', _lib.MheDesc)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "hilo_hip.h"', 'int main(void) {']
    for cname, cls in pairs:
        body = re.search(r'typedef struct %s \{(.*?)\} %s;' % (cname, cname), header, re.S).group(1)
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        fields = []
        for decl in body.split(';'):
            decl = decl.strip()
            if decl:                                      # `type a, b, *c` declares several fields
                for part in decl.split(','):
                    fields.append(re.sub(r'\[.*\]', '', part.replace('*', ' ').split()[-1]))
        assert fields == [f[0] for f in cls._fields_], (cname, fields, [f[0] for f in cls._fields_])
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        lines += [f'  printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));' for f in fields]
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).strip().split('\n'))
    for cname, cls in pairs:
        assert int(got[cname]) == C.sizeof(cls), cname
        for f, _ in cls._fields_:
            assert int(got[f'{cname}.{f}']) == getattr(cls, f).offset, (cname, f)


def test_opcode_tables_match_the_header():
    """Every `HILO_X_*` (expression VM), `HILO_K_*` (kernels) and `HILO_M_*` (means) opcode of the header has the same value
    in the host-side compilers (hilo_mpc_amd/expr.py, hilo_mpc_amd/gp.py)."""
    import re
    from hilo_mpc_amd import expr, gp
    header = open(os.path.join(ROOT, 'include', 'hilo_hip.h')).read()
    seen = 0
    for prefix, mod in (('X', expr), ('K', gp), ('M', gp)):
        for name, value in re.findall(r'#define HILO_%s_(\w+)\s+(\(?-?\d+\)?)' % prefix, header):
            assert getattr(mod, f'{prefix}_{name}') == int(value.strip('()')), (prefix, name)
            seen += 1
    assert seen >= 16 + 11 + 6
