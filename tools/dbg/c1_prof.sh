#!/bin/bash
# Debug aid: rocprofv3 kernel durations of the C1 bench line (B = 1024 closed loop and the single instance inside finish())
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
d=gpurun_out/c1prof; rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python bench.py --config C1 --no-cpu-baseline --steps 30 > $d.log 2>&1
grep '^{"metric"' $d.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('line:', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['single_instance_latency_us'])"
python - "$(find $d -name '*kernel_stats.csv' | head -1)" "$(find $d -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print('   ', r['Name'][:70], r['Calls'], 'avg ns', r['AverageNs'], 'min', r['MinNs'], 'max', r['MaxNs'])
rows = [r for r in csv.DictReader(open(sys.argv[2])) if 'qp_' in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp']), r.get('Grid_Size', r.get('Grid_Size_X'))) for r in rows]
print('    qp launches (ns, grid):', d[:3], '...', d[-6:])
PY
