#!/usr/bin/env python3
"""Condense the issue-counter passes of profiles/run_pmc_valu.sh into profiles/<tag>_pmc_issue.json: per counter the mean over
the steady-state launches (the last 6 of the 20 launches of each pass) of the solve kernel, summed over the launch's waves.

    python profiles/summarize_pmc.py gpurun_out/pmc_r01 r01
"""
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
KEY = 'ocp_solve_kernel'
out = {}
for p in sorted(glob.glob(os.path.join(src, '*', f'{tag}_counter_collection.csv'))):
    per = {}
    for r in csv.DictReader(open(p)):
        if KEY in r['Kernel_Name']:
            per.setdefault(r['Counter_Name'], {}).setdefault(r['Dispatch_Id'], 0.0)
            per[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    for name, d in per.items():
        v = [d[k] for k in sorted(d, key=int)]
        out[name] = {'n': len(v), 'steady_mean': sum(v[-6:]) / len(v[-6:])}
# the GP prediction kernel's matrix-core counters (own pass, run_pmc_valu.sh)
gp = {}
for p in sorted(glob.glob(os.path.join(src, 'gp', '*', f'{tag}_counter_collection.csv'))):
    per = {}
    for r in csv.DictReader(open(p)):
        if 'gp_predict' in r['Kernel_Name']:
            per.setdefault(r['Counter_Name'], {}).setdefault(r['Dispatch_Id'], 0.0)
            per[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    for name, d in per.items():
        v = [d[k] for k in sorted(d, key=int)]
        gp[name] = {'n': len(v), 'steady_mean': sum(v[-4:]) / len(v[-4:])}
if gp:
    out['gp_predict_reg_kernel'] = gp
here = os.path.dirname(os.path.abspath(__file__))
json.dump(out, open(os.path.join(here, f'{tag}_pmc_issue.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
