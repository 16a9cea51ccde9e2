"""Fill the run-time compiler's cache (hilo_mpc_amd/jit_cache, or HILO_JIT_CACHE) on a machine WITHOUT a GPU.

With HILO_JIT_COMPILE_ONLY=1 every `setup()` of a run-time compiled problem compiles its kernels with hiprtc and stops
(hilo_nmpc_create / hilo_kf_create return HILO_COMPILED_ONLY).  This script drives the GPU tests' problem constructions in that
mode - the tests themselves fail at their first device call, which is expected and ignored - so that the code objects travel
with the tree and the GPU box does not spend its first minutes in the compiler.  Problems that need a trained GP on the device
before setup() are not reached.

    python tools/warm_jit_cache.py [pytest selection ...]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(argv):
    cache = os.environ.get('HILO_JIT_CACHE') or os.path.join(ROOT, 'hilo_mpc_amd', 'jit_cache')
    before = set(os.listdir(cache)) if os.path.isdir(cache) else set()
    env = dict(os.environ, HILO_JIT_COMPILE_ONLY='1')
    sel = argv or ['tests']
    jobs = os.environ.get('HILO_WARM_JOBS', str(max(1, (os.cpu_count() or 2) - 2)))      # pytest-xdist workers (compilations are independent)
    subprocess.call([sys.executable, '-m', 'pytest', '-m', 'gpu', '-q', '-p', 'no:cacheprovider', '--tb=no', '--no-header',
                     '-W', 'ignore', '-n', jobs] + sel, cwd=ROOT, env=env, stdout=subprocess.DEVNULL)
    if not argv:
        subprocess.call([sys.executable, os.path.abspath(__file__), '--learned-filter'], cwd=ROOT, env=env)
    after = set(os.listdir(cache)) if os.path.isdir(cache) else set()
    print(f"{len(after - before)} new code object(s), {len(after)} in {cache}")
    # the backend's EXEC-prologue defect shows up in run-time compiled problems as well: check what was compiled
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import check_exec_prologue as guard
    bad = [f for f in sorted(after) if f.endswith('.hsaco') and os.path.exists(guard.OBJDUMP) and guard.main(os.path.join(cache, f)) != 0]
    print("EXEC-prologue guard:", "clean" if not bad else f"{len(bad)} object(s) flagged: {bad}")
    return 1 if bad else 0


def learned_filter():
    """The filter kernels of tests/test_kf_learned_gpu.py: the source only depends on the model's expressions and on the KIND of the
    learned term (squared exponential over two states), so an untrained GP object stands in for the trained one."""
    sys.path.insert(0, ROOT)
    from hilo_mpc_amd import EKF, GP, Kernel
    from tests.problems import symbolic_model
    gp = GP(['S', 'I'], ['mu'], kernel=Kernel.squared_exponential(active_dims=[0, 1], length_scales=[8., 1.5], ard=True),
            noise_variance=1e-3)
    gp._handle = 1                                   # never dereferenced in compile-only mode
    try:
        m = symbolic_model('chemostat4_mu')
        m.substitute_from(gp)
        EKF(m.discretize('erk', order=4).setup(dt=1.)).setup()
    finally:
        gp._handle = None                            # (nothing to destroy)


if __name__ == '__main__':
    if sys.argv[1:] == ['--learned-filter']:
        learned_filter()
        sys.exit(0)
    sys.exit(main(sys.argv[1:]))
