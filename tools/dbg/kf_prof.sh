#!/bin/bash
# Debug aid: rocprofv3 kernel durations of the filter bench lines (EKF / UKF at the BASELINE batch, K = 16 and K = 1)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for c in C3-ekf C3-ukf; do for k in 16 1; do
  d=gpurun_out/kfprof_${c}_$k; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python bench.py --config $c --no-cpu-baseline --steps 30 --kf-steps $k > $d.log 2>&1
  grep '^{"metric"' $d.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c K=$k line:', d['roofline']['kernel_ms'], d['value'])"
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:4]:
    print('   ', r['Name'][:60], r['Calls'], 'avg ns', r['AverageNs'], 'min', r['MinNs'], 'max', r['MaxNs'])
PY
done; done
