"""Guard against a code-generation defect of the ROCm 7.2 AMDGPU backend that silently corrupts registers.

When a divergent loop or if-region ends, the lanes are re-enabled by an `s_or_b64 exec, exec, sN` at the top of the join
block.  Under register pressure the register allocator sometimes places a live-range-split copy or a spill
(`v_mov_b32 vA, vB`, `v_accvgpr_write_b32`, `scratch_store`) in that block BEFORE the `s_or_b64`.  The block is entered
through `s_cbranch_execz` (or by falling out of an `s_cbranch_execnz` loop), i.e. with EXEC == 0, so the copy executes for
no lane and the value read back later is garbage.  It cost this project the filter size of the constrained NMPC variants
(DESIGN.md 5.1); the solver pins its wave-uniform state to scalar registers since, and this script checks what is left:

    python tools/check_exec_prologue.py [libhilo_hip.so]

disassembles every gfx950 code object of the library and reports join blocks entered with EXEC == 0 whose code before the
`s_or_b64 exec` consists of scalar / lane instructions and at least one vector copy or spill.  Exit status 1 when there are any."""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
OBJCOPY = '/opt/rocm/lib/llvm/bin/llvm-objcopy'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
LANE_OPS = ('v_writelane', 'v_readlane', 'v_readfirstlane')      # do not depend on EXEC
COPY = re.compile(r'^(v_mov_b32|v_mov_b64|v_accvgpr_write|v_accvgpr_read|v_accvgpr_mov|scratch_store|scratch_load)')


def code_objects(lib, tmp):
    fat = os.path.join(tmp, 'fat.bin')
    subprocess.check_call([OBJCOPY, '-O', 'binary', '--only-section=.hip_fatbin', lib, fat])
    d = open(fat, 'rb').read()
    out, pos = [], 0
    while True:
        i = d.find(MAGIC, pos)
        if i < 0:
            return out
        nb = struct.unpack_from('<Q', d, i + 24)[0]
        off = i + 32
        for _ in range(nb):
            o, sz, tl = struct.unpack_from('<QQQ', d, off)
            off += 24
            triple = d[off:off + tl].decode()
            off += tl
            if 'gfx950' in triple and sz:
                p = os.path.join(tmp, f'co_{len(out)}.elf')
                open(p, 'wb').write(d[i + o:i + o + sz])
                out.append(p)
        pos = i + 24


LINE = re.compile(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]{12}):')


def check(elf):
    txt = subprocess.run([OBJDUMP, '-d', elf], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    ins, func = [], None                                  # (addr, mnemonic, operands, function)
    for ln in txt.split('\n'):
        m = re.match(r'^[0-9a-f]+ <(\S+)>:', ln)
        if m:
            func = m.group(1)
            continue
        m = LINE.match(ln)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2), func))
    index = {a: k for k, (a, *_) in enumerate(ins)}
    starts = set()
    for k, (a, op, args, _) in enumerate(ins):
        if op == 's_cbranch_execz':
            off = int(args.split()[0])
            starts.add(index.get(a + 4 + 4 * (off - 65536 if off >= 32768 else off)))   # simm16 is printed unsigned
        elif op == 's_cbranch_execnz' and k + 1 < len(ins):
            starts.add(k + 1)                             # loop left with every lane masked off
    hits = []
    for k in sorted(s for s in starts if s is not None):
        bad = []
        for a, op, args, f in ins[k:k + 64]:
            if op == 's_or_b64' and args.replace(' ', '').startswith('exec,exec,'):
                if bad:
                    hits.append((ins[k][3], bad))         # only copies / spills / scalar code before the lanes come back
                break
            if op.startswith('s_') and not op.startswith(('s_cbranch', 's_branch', 's_setpc', 's_swappc', 's_endpgm')) \
                    and not re.match(r'exec', args):
                continue
            if op.startswith(LANE_OPS):
                continue
            if COPY.match(op):
                bad.append(f'{a:#x}: {op} {args}')
                continue
            break                                          # real work: the tail of a predicated region, not a prologue
    return hits


def main(lib):
    total = 0
    with tempfile.TemporaryDirectory() as tmp:
        # a run-time compiled code object (jit_cache/*.hsaco) is a plain gfx950 ELF, the library a fat binary of several
        for elf in ([lib] if lib.endswith('.hsaco') else code_objects(lib, tmp)):
            for func, bad in check(elf):
                total += 1
                print(f'{func}:')
                for b in bad[:6]:
                    print('    ' + b)
    print(f'{total} join-block prologue(s) with a vector copy before the EXEC restore')
    return 1 if total else 0


if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, '..', 'hilo_mpc_amd', 'libhilo_hip.so')))
