#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_rNN/<config>/{trace,pmc_fetch,pmc_write}, written by profiles/run_profile.sh)
into the small, tracked summaries under profiles/: per configuration the --stats table, the launch durations of the dominant
kernel, the PMC byte counters and the bench line of the same build.

    python profiles/summarize.py gpurun_out/prof_r02 r02
"""
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
# substring that identifies the dominant kernel of each configuration in the trace
KEYS = {'C1': 'qp_ocp_kernel', 'C2': 'ocp_solve_kernel', 'C3-mhe': 'ocp_solve_kernel', 'C3-ekf': 'kf_team_kernel',
        'C3-ukf': 'kf_team_kernel', 'C4': 'ocp_solve_kernel', 'C5': 'hilo_user_solve', 'C5-dae': 'hilo_user_solve', 'gp-predict': 'gp_predict'}


def find(d, pattern):
    f = glob.glob(os.path.join(d, '**', pattern), recursive=True)
    return f[0] if f else None


for cdir in sorted(glob.glob(os.path.join(src, '*'))):
    cfg = os.path.basename(cdir)
    if cfg not in KEYS:
        continue
    key = KEYS[cfg]
    out = {'tag': tag, 'config': cfg, 'kernel_key': key}
    line = None
    bj = os.path.join(cdir, 'bench_n1.json')
    if os.path.exists(bj):
        for ln in open(bj):
            if ln.startswith('{"metric"'):
                line = json.loads(ln)
    if line:
        out['bench_line'] = line
        timed = line['steps']
    else:
        timed = 20
    stats = find(os.path.join(cdir, 'trace'), '*kernel_stats.csv')
    if stats:
        rows = list(csv.DictReader(open(stats)))
        out['kernel_stats_top'] = [{k: (r[k][:100] if k == 'Name' else r[k]) for k in ('Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage')}
                                   for r in rows[:8]]
        k = [r for r in rows if key in r['Name']]
        if k:
            k = k[0]
            out['kernel_name'] = k['Name'][:160]
            out['stats_all_launches'] = {'calls': int(k['Calls']), 'avg_ns': float(k['AverageNs']), 'min_ns': float(k['MinNs']),
                                         'max_ns': float(k['MaxNs']), 'percentage': float(k['Percentage'])}
    trace = find(os.path.join(cdir, 'trace'), '*kernel_trace.csv')
    if trace:
        rows = [r for r in csv.DictReader(open(trace)) if key in r['Kernel_Name']]
        if rows and 'Grid_Size_X' in rows[0]:   # the step's launches (full batch), not the single-instance latency probe of C1
            gmax = max(int(r['Grid_Size_X']) for r in rows)
            rows = [r for r in rows if int(r['Grid_Size_X']) == gmax]
        d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]
        if d:
            # the headline configuration continues its loop for 50 more steps after the timed region (bench.py `settled_loop`): the
            # timed region is the `timed` launches in front of those
            tail = 50 if (line and 'settled_loop' in line and len(d) >= timed + 50) else 0
            t = d[len(d) - tail - timed:len(d) - tail]
            out['timed_region'] = {'launches': len(t), 'avg_ns': sum(t) / len(t), 'min_ns': min(t), 'max_ns': max(t)}
            if tail:
                ts = d[-tail:]
                out['settled_region'] = {'launches': len(ts), 'avg_ns': sum(ts) / len(ts), 'min_ns': min(ts), 'max_ns': max(ts)}
            r0 = rows[-1]
            out['resources'] = {k: r0.get(k) for k in ('Workgroup_Size_X', 'Grid_Size_X', 'LDS_Block_Size', 'Scratch_Size', 'VGPR_Count',
                                                       'Accum_VGPR_Count', 'SGPR_Count') if k in r0}
    for name, ctr in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
        p = find(os.path.join(cdir, name), '*counter_collection.csv')
        if p:
            v = [float(r['Counter_Value']) for r in csv.DictReader(open(p)) if key in r['Kernel_Name'] and r['Counter_Name'] == ctr]
            if v:
                out[ctr + '_KB_per_launch'] = {'n': len(v), 'mean': sum(v) / len(v), 'min': min(v), 'max': max(v),
                                               'warm_launches_mean': sum(v[2:]) / max(1, len(v[2:]))}
    if line and 'FETCH_SIZE_KB_per_launch' in out and 'WRITE_SIZE_KB_per_launch' in out:
        # the line was printed before these passes were condensed: its `traffic` field is filled from THIS run's counters
        line['roofline']['traffic'] = (out['FETCH_SIZE_KB_per_launch']['warm_launches_mean'] +
                                       out['WRITE_SIZE_KB_per_launch']['warm_launches_mean']) * 1024
        line['roofline']['traffic_source'] = f'profiles/{tag}_{cfg}_summary.json (FETCH_SIZE + WRITE_SIZE passes of this run)'
    with open(os.path.join(here, f'{tag}_{cfg}_summary.json'), 'w') as f:
        json.dump(out, f, indent=1)
    brief = {k: v for k, v in out.items() if k in ('config', 'kernel_name', 'timed_region', 'FETCH_SIZE_KB_per_launch', 'WRITE_SIZE_KB_per_launch')}
    if line:
        brief['value'] = line['value']
        brief['roofline'] = {k: line['roofline'].get(k) for k in ('bound', 'achieved', 'frac', 'kernel_ms')}
    print(json.dumps(brief))
