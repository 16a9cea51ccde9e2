"""In-tree build of libhilo_hip.so (hipcc, gfx950 only).  Used by __graft_entry__.build() and by developers;
never invoked implicitly at import time - a missing library is a hard error (see _lib.py)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libhilo_hip.so')
SOURCES = ['hilo_api.hip', 'hilo_kf.hip', 'hilo_gp.hip', 'hilo_nmpc.hip', 'hilo_qp.hip', 'hilo_mhe.hip',
           'hilo_nmpc_gen_chemostat4.hip', 'hilo_nmpc_gen_robot6.hip', 'hilo_nmpc_coll.hip', 'hilo_nmpc_tv.hip', 'hilo_mhe_est.hip', 'hilo_nmpc_long.hip', 'hilo_jit.hip', 'hilo_nmpc_user.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wno-unused-result'] + \
    os.environ.get('HILO_EXTRA_FLAGS', '').split()      # developer knob (tuning sweeps), empty in normal builds
# Translation units built around the interior-point engine (csrc/hilo_ocp.h): the f64 matrix-core products of its Riccati stage take
# and leave their tiles in VGPRs (no v_accvgpr_read / _write around every product; hiprtc's compiler does not know the option, hilo_jit.hip)
OCP_FLAGS = ['-mllvm', '-amdgpu-mfma-vgpr-form=1']
OCP_UNITS = {'hilo_nmpc.hip', 'hilo_mhe.hip', 'hilo_nmpc_gen_chemostat4.hip', 'hilo_nmpc_gen_robot6.hip', 'hilo_nmpc_coll.hip',
             'hilo_nmpc_tv.hip', 'hilo_mhe_est.hip', 'hilo_nmpc_long.hip'}


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, jobs=None, tag=None, extra_flags=()):
    """Compile every .hip translation unit to an object (in parallel) and link the shared library.
    tag / extra_flags: a developer variant (tuning sweeps, debug counters) next to the product build - objects in build_<tag>/,
    library libhilo_hip_<tag>.so, loaded instead of the product library when HILO_LIB_PATH points at it."""
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(HERE), 'include', 'hilo_hip.h'))
    objdir = os.path.join(HERE, 'build' if not tag else f'build_{tag}')
    lib = LIB if not tag else os.path.join(HERE, f'libhilo_hip_{tag}.so')
    flags = FLAGS + list(extra_flags)
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src).replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + flags + (OCP_FLAGS if os.path.basename(src) in OCP_UNITS else []) + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors='replace'))
            raise RuntimeError(f'hipcc failed on {src}')
    if force or procs or _stale(lib, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs + ['-lhiprtc', '-ldl']   # hiprtc: run-time compiled user models (csrc/hilo_jit.hip)
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == '__main__':
    build(force='--force' in sys.argv)
