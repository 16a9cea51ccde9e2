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


def test_round_2_profiles_are_consistent_with_their_bench_lines():
    """One summary per configuration (profiles/r02_<config>_summary.json: rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes +
    the bench line of the same build and run): the HIP-event duration of the line agrees with the trace, `traffic` is the sum of
    the two PMC passes, the fractions follow from achieved / peak, every line carries the CPU baseline."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r02_*_summary.json')))
    assert {os.path.basename(f)[4:-13] for f in files} >= {'C1', 'C2', 'C3-mhe', 'C3-ekf', 'C3-ukf', 'C4', 'C5', 'gp-predict'}
    for f in files:
        s = json.load(open(f))
        line, cfg = s['bench_line'], s['config']
        r = line['roofline']
        trace_ms = s['timed_region']['avg_ns'] * 1e-6
        if cfg in ('C3-ekf', 'C3-ukf'):
            # a 7 - 15 us kernel: the event pair also sees the launch gap, the trace only the kernel
            assert trace_ms <= r['kernel_ms'] < trace_ms + 0.05, cfg
        else:
            assert abs(trace_ms - r['kernel_ms']) / trace_ms < 0.08, (cfg, trace_ms, r['kernel_ms'])
        traffic = (s['FETCH_SIZE_KB_per_launch']['warm_launches_mean'] + s['WRITE_SIZE_KB_per_launch']['warm_launches_mean']) * 1024
        assert abs(r['traffic'] - traffic) <= 1e-9 * traffic, cfg
        assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and r['bound'] in ('hbm', 'mfma'), cfg
        assert set(line['cpu_baseline']) >= {'value', 'unit', 'cores', 'kind', 'sample'}, cfg
        assert line['config']['name'] == cfg and line['n_gpus'] == 1 and line['vs_baseline'] is None


def test_round_3_profiles_are_consistent_with_their_bench_lines():
    """profiles/r03_<config>_summary.json, same checks as for round 2.  C1 and the filter lines bracket a whole Python step with
    their HIP events (the solve plus the small kernels around it): the event duration is an upper bound of the trace's."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r03_*_summary.json')))
    assert {os.path.basename(f)[4:-13] for f in files} >= {'C1', 'C2', 'C3-mhe', 'C3-ekf', 'C3-ukf', 'C4', 'C5', 'gp-predict'}
    for f in files:
        s = json.load(open(f))
        line, cfg = s['bench_line'], s['config']
        r = line['roofline']
        trace_ms = s['timed_region']['avg_ns'] * 1e-6
        if cfg in ('C1', 'C3-ekf', 'C3-ukf'):
            assert trace_ms <= r['kernel_ms'] < trace_ms + 0.1, cfg
        else:
            assert abs(trace_ms - r['kernel_ms']) / trace_ms < 0.08, (cfg, trace_ms, r['kernel_ms'])
        traffic = (s['FETCH_SIZE_KB_per_launch']['warm_launches_mean'] + s['WRITE_SIZE_KB_per_launch']['warm_launches_mean']) * 1024
        assert abs(r['traffic'] - traffic) <= 1e-9 * traffic, cfg
        assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and r['bound'] in ('hbm', 'mfma'), cfg
        cb = line['cpu_baseline']
        assert set(cb) >= {'value', 'unit', 'cores', 'kind', 'sample'}, cfg
        # every line carries a C++ / OpenMP leg: true core count, one-core figure next to it
        assert cb['cores'] >= 1 and cb['one_core_value'] > 0 and 'C++17' in cb['sample'], cfg
        assert line['config']['name'] == cfg and line['n_gpus'] == 1 and line['vs_baseline'] is None


def test_cpu_baseline_leg_of_c2_runs_here():
    """The C++/OpenMP leg of the C2 line on a reduced sample (no GPU needed): fields, and all-core >= one-core."""
    from tests.problems import C2
    b = _bench()
    out = b.cpu_baseline_c2(C2, nst=3, slsqp=False)
    assert out['kind'] == 'port' and out['cores'] >= 1 and out['value'] > 0 and out['one_core_value'] > 0
    assert out['frac_status_1_or_2'] == 1.0


def test_cpu_baseline_legs_of_the_filters_and_the_qp_run_here():
    """The C++/OpenMP legs of the C3 filter lines and the C1 line on a reduced sample (no GPU needed)."""
    b = _bench()
    for kind in ('ekf', 'ukf'):
        out = b.cpu_leg_kf(kind, 4, budget=.3, min_batch=256)
        assert out['kind'] == 'port' and out['cores'] >= 1 and out['value'] > 0 and out['one_core_value'] > 0 and out['unit'] == 'steps/s'
    # the QP leg runs the GPU line's own closed loop (same draw of measured states, same warm-up + timed steps): the status fraction
    # and the iteration count of its timed steps are those of `profiles/r06_C1_summary.json`'s GPU line (0.946, 4.31)
    out = b.cpu_leg_qp(budget=.3, min_batch=128, warmup=5, steps=20)
    assert out['value'] > 0 and 0.9 < out['frac_status_1'] < 1.0 and 3.5 < out['mean_qp_iters'] < 5.5


def test_cpu_baseline_leg_of_the_estimator_runs_here():
    """The C++/OpenMP leg of the C3 MHE line on a reduced sample (no GPU needed)."""
    from tests.problems import C3B
    out = _bench().cpu_leg_mhe(C3B, nst=2, per_thread=1)
    assert out['kind'] == 'port' and out['cores'] >= 1 and out['value'] > 0 and out['one_core_value'] > 0 and out['unit'] == 'steps/s'
    assert out['frac_status_1_or_2'] == 1.0 and 5 < out['mean_ipm_iters'] < 40


def test_cpu_baseline_leg_of_c5_runs_here():
    """The C++/OpenMP leg of the C5 line on a reduced sample (no GPU needed)."""
    from tests.problems import C5
    out = _bench().cpu_leg_c5(dict(C5, N=12), nst=2, per_thread=1)
    assert out['kind'] == 'port' and out['cores'] >= 1 and out['value'] > 0 and out['one_core_value'] > 0 and out['unit'] == 'steps/s'
    assert out['frac_status_1_or_2'] == 1.0 and 3 < out['mean_ipm_iters'] < 60


def test_cpu_baseline_leg_of_the_prediction_runs_here():
    import numpy as np
    out = _bench().cpu_leg_gp(np.stack([np.linspace(0, 40, 512), np.linspace(0, 4, 512)]), budget=.2)
    assert out['kind'] == 'port' and out['value'] > 0 and out['one_core_value'] > 0 and out['unit'] == 'predictions/s'


def test_bench_spawns_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` without a torchrun environment re-executes itself under torch.distributed.run with N processes
    on 127.0.0.1 (the driver's own launch line) and hands its exit status through."""
    import argparse
    import sys
    b = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 7
    monkeypatch.setattr(b.subprocess, 'call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--config', 'C4'])
    rc = b.respawn(argparse.Namespace(gpus=4))
    cmd = seen['cmd']
    assert rc == 7 and cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nnodes=1' in cmd and '--nproc-per-node=4' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    assert cmd[-6:] == ['--gpus', '4', '--steps', '3', '--config', 'C4'] and cmd[-7].endswith('bench.py')
    assert seen['env'].get('HSA_ENABLE_IPC_MODE_LEGACY') == '0'


def test_committed_counters_are_withheld_when_the_profiled_kernel_is_another_one():
    """bench.py carries HBM traffic / issue counters from COMMITTED rocprofv3 passes (they cannot be collected inside the timed run).
    When this run's kernel time differs from the profiled kernel's by more than the band, the counters are withheld and the line says
    why - a later kernel change does not keep stale counters on the driver's line."""
    b = _bench()
    import json
    import os
    prof = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r06_C2_summary.json')))
    ms = prof['timed_region']['avg_ns'] * 1e-6
    t, src = b.pmc_traffic_bytes('C2', with_source=True, kernel_ms=ms)
    assert t and t > 0 and 'r06_C2_summary.json' in src and 'withheld' not in src
    t, src = b.pmc_traffic_bytes('C2', with_source=True, kernel_ms=ms * 2.)
    assert t is None and 'withheld' in src
    iss, src = b.pmc_issue('C2', kernel_ms=ms * .4)
    assert iss is None and 'withheld' in src
    assert b._stale(None, 1.) is None and b._stale(1e6, 1.) is None and b._stale(1e6, 1.2) is None and 'withheld' in b._stale(1e6, 1.5)
