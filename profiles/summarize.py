#!/usr/bin/env python3
"""Condense a rocprofv3 output directory (gpurun_out/prof_rNN/{trace,pmc_fetch,pmc_write}) into the small, tracked
summaries under profiles/: the --stats table, per-launch durations of the dominant kernel, PMC byte counters.

    python profiles/summarize.py gpurun_out/prof_r01 r01 [timed_launches]
"""
import csv
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
timed = int(sys.argv[3]) if len(sys.argv) > 3 else 20
here = os.path.dirname(os.path.abspath(__file__))
KEY = sys.argv[4] if len(sys.argv) > 4 else 'ocp_solve_kernel'

out = {'tag': tag, 'kernel': KEY}
stats = os.path.join(src, 'trace', f'{tag}_kernel_stats.csv')
if os.path.exists(stats):
    rows = list(csv.DictReader(open(stats)))
    short = []
    for r in rows:
        r = dict(r)
        r['Name'] = r['Name'][:110]
        short.append(r)
    with open(os.path.join(here, f'{tag}_kernel_stats.csv'), 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=list(short[0].keys()))
        w.writeheader()
        w.writerows(short)
    k = [r for r in rows if KEY in r['Name']][0]
    out['stats_all_launches'] = {'calls': int(k['Calls']), 'avg_ns': float(k['AverageNs']),
                                 'min_ns': float(k['MinNs']), 'max_ns': float(k['MaxNs']),
                                 'percentage': float(k['Percentage'])}
trace = os.path.join(src, 'trace', f'{tag}_kernel_trace.csv')
if os.path.exists(trace):
    rows = [r for r in csv.DictReader(open(trace)) if KEY in r['Kernel_Name']]
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]
    out['launch_ns'] = d
    t = d[-timed:]
    out['timed_region'] = {'launches': len(t), 'avg_ns': sum(t) / len(t), 'min_ns': min(t), 'max_ns': max(t)}
    r0 = rows[-1]
    out['resources'] = {k: r0.get(k) for k in ('Workgroup_Size', 'Grid_Size', 'LDS_Block_Size', 'Scratch_Size',
                                               'VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count') if k in r0}
for name, ctr in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    p = os.path.join(src, name, f'{tag}_counter_collection.csv')
    if os.path.exists(p):
        v = [float(r['Counter_Value']) for r in csv.DictReader(open(p)) if KEY in r['Kernel_Name'] and r['Counter_Name'] == ctr]
        if v:
            out[ctr + '_KB_per_launch'] = {'n': len(v), 'mean': sum(v) / len(v), 'min': min(v), 'max': max(v),
                                           'warm_launches_mean': sum(v[2:]) / max(1, len(v[2:]))}
log = os.path.join(src, 'bench_under_rocprof.log')
if os.path.exists(log):
    for line in open(log):
        if line.startswith('{"metric"'):
            out['bench_line_under_profiler'] = json.loads(line)
with open(os.path.join(here, f'{tag}_summary.json'), 'w') as f:
    json.dump(out, f, indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ('launch_ns', 'bench_line_under_profiler')}, indent=1))
