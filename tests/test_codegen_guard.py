"""Build guard: no vector copy / spill may sit in front of the EXEC restore of a join block (tools/check_exec_prologue.py).
The ROCm 7.2 backend produced that for two NMPC variants under register pressure; the copy executes for no lane and the
solver's state is garbage afterwards.  The check disassembles the shipped library, so it covers every kernel."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import check_exec_prologue as guard                                                             # noqa: E402

LIB = os.path.join(ROOT, 'hilo_mpc_amd', 'libhilo_hip.so')


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(guard.OBJDUMP)), reason='library or llvm-objdump missing')
def test_no_vector_copy_before_exec_restore(capsys):
    assert guard.main(LIB) == 0, capsys.readouterr().out


@pytest.mark.skipif(not os.path.exists(guard.OBJDUMP), reason='llvm-objdump missing')
def test_run_time_compiled_problems_in_the_cache_are_clean(capsys):
    """The defect shows up in run-time compiled problems too (round 3: behind the exits of lane-dependent loops in the collocation
    Newton iteration and in the index decomposition of the Taylor directions): every code object that travels with the tree is
    checked.  (An empty cache - a fresh clone - has nothing to check.)"""
    cache = os.path.join(ROOT, 'hilo_mpc_amd', 'jit_cache')
    objs = sorted(f for f in os.listdir(cache) if f.endswith('.hsaco')) if os.path.isdir(cache) else []
    bad = [f for f in objs if guard.main(os.path.join(cache, f)) != 0]
    assert not bad, (bad, capsys.readouterr().out[-2000:])


def test_guard_recognises_the_pattern(tmp_path):
    """The checker on a synthetic listing: a copy before `s_or_b64 exec` at the target of an `s_cbranch_execz` is reported,
    the same copy after it is not."""
    def listing(body):
        lines = ['0000000000001000 <k>:', '\ts_and_saveexec_b64 s[0:1], vcc // 000000001000: AAAAAAAA',
                 '\ts_cbranch_execz 1 // 000000001004: BF880001', '\tv_add_f64 v[0:1], v[0:1], v[2:3] // 000000001008: AAAAAAAA']
        a = 0x100c
        for ins in body:
            lines.append(f'\t{ins} // {a:012X}: AAAAAAAA')
            a += 4
        return '\n'.join(lines) + '\n'
    bad = listing(['v_writelane_b32 v254, s36, 43', 'v_mov_b32_e32 v189, v38', 's_or_b64 exec, exec, s[0:1]', 's_endpgm'])
    good = listing(['v_writelane_b32 v254, s36, 43', 's_or_b64 exec, exec, s[0:1]', 'v_mov_b32_e32 v189, v38', 's_endpgm'])
    import subprocess
    real = subprocess.run

    def fake(text):
        def run(cmd, **kw):
            class R:
                stdout = text
            return R()
        return run
    try:
        subprocess.run = fake(bad)
        assert len(guard.check('x')) == 1
        subprocess.run = fake(good)
        assert guard.check('x') == []
    finally:
        subprocess.run = real
