#!/usr/bin/env python3
"""Condense the issue-counter passes of profiles/run_pmc_valu.sh into profiles/<tag>_pmc_issue[_<config>].json: per counter the mean
over the steady-state launches (the last `steps` launches of each pass) of the configuration's solve kernel, summed over the
launch's waves; the kernel's duration in shader clocks from the kernel trace of the same passes; and the derived fractions the
bench line carries (`mfma_busy_frac` = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration), `valu_busy_frac` = SQ_ACTIVE_INST_VALU /
SQ_WAVE_CYCLES, instructions per wave).

    python profiles/summarize_pmc.py gpurun_out/pmc_r05 r05            (C2 + the GP prediction kernel)
    python profiles/summarize_pmc.py gpurun_out/pmc_r05_C4 r05 C4
"""
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
cfg = sys.argv[3] if len(sys.argv) > 3 else 'C2'
KEY = os.environ.get('PMC_KEY') or ('hilo_user_solve' if cfg in ('C5', 'C5-dae') else 'ocp_solve_kernel')      # cfg 'icache': the I-cache passes of C2 (run_pmc_icache.sh)
CLOCK_GHZ = 2.4          # MI355X shader clock (MI355X_MICROARCH.md)
N_SIMD = 1024
STEADY = {'C2': 6, 'icache': 6, 'C4': 4, 'C3-mhe': 4, 'C5': 3, 'C5-dae': 2}.get(cfg, 4)     # timed launches of each pass (run_round.sh)


def collect(pattern, key, steady):
    out, dur = {}, []
    for p in sorted(glob.glob(pattern)):
        per = {}
        for r in csv.DictReader(open(p)):
            if key in r['Kernel_Name']:
                per.setdefault(r['Counter_Name'], {}).setdefault(r['Dispatch_Id'], 0.0)
                per[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
        for name, d in per.items():
            v = [d[k] for k in sorted(d, key=int)]
            out[name] = {'n': len(v), 'steady_mean': sum(v[-steady:]) / len(v[-steady:])}
        kt = p.replace('counter_collection.csv', 'kernel_trace.csv')
        if os.path.exists(kt):
            rows = [r for r in csv.DictReader(open(kt)) if key in r['Kernel_Name']]
            rows.sort(key=lambda r: int(r['Dispatch_Id']))
            d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows[-steady:]]
            if d:
                dur.append(sum(d) / len(d))
    if dur:
        out['kernel_ns_under_pmc'] = sum(dur) / len(dur)
        out['kernel_cycles'] = out['kernel_ns_under_pmc'] * CLOCK_GHZ
    g = lambda k: out[k]['steady_mean'] if k in out else None     # noqa: E731
    if g('SQ_VALU_MFMA_BUSY_CYCLES') is not None and 'kernel_cycles' in out:
        out['mfma_busy_frac'] = g('SQ_VALU_MFMA_BUSY_CYCLES') / (N_SIMD * out['kernel_cycles'])
    if g('SQ_ACTIVE_INST_VALU') and g('SQ_WAVE_CYCLES'):
        out['valu_busy_frac'] = g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES')
    if g('SQ_INSTS_VALU') and 'kernel_cycles' in out:     # a wave's fp64 / fp32 instruction occupies its SIMD for 4 clocks
        out['valu_issue_floor_frac'] = 4.0 * g('SQ_INSTS_VALU') / (N_SIMD * out['kernel_cycles'])
    if g('SQ_WAIT_INST_LDS') is not None and g('SQ_WAVE_CYCLES'):
        out['lds_wait_frac'] = g('SQ_WAIT_INST_LDS') / g('SQ_WAVE_CYCLES')
    if g('SQC_ICACHE_REQ') and g('SQC_ICACHE_MISSES') is not None:
        out['icache_miss_frac'] = g('SQC_ICACHE_MISSES') / g('SQC_ICACHE_REQ')
    return out


out = collect(os.path.join(src, '*', f'{tag}_counter_collection.csv'), KEY, STEADY)
gp = collect(os.path.join(src, 'gp', '*', f'{tag}_counter_collection.csv'), 'gp_predict', 4)
if gp:
    out['gp_predict_reg_kernel'] = gp
here = os.path.dirname(os.path.abspath(__file__))
name = f'{tag}_pmc_issue.json' if cfg == 'C2' else f'{tag}_pmc_issue_{cfg}.json'
json.dump(out, open(os.path.join(here, name), 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if not isinstance(v, dict) or k == 'gp_predict_reg_kernel'}, indent=1))
