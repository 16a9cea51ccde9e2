#!/bin/bash
# GPU box: selected test files + selected bench configs.   gpurun -- bash tools/gpu/run_sel.sh <tag> "<pytest args>" [bench configs ...]
TAG=${1:-r3}; SEL=$2; shift; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest $SEL -m gpu -q --maxfail=20 -p no:cacheprovider -s > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|hiprtc seconds|rc=|Error" $OUT/pytest.log | tail -n 30
for cfg in "$@"; do
  timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err; cut -c1-700 $OUT/bench_$cfg.json; tail -n 3 $OUT/bench_$cfg.err
done
timeout 300 python tools/phase_profile_c5.py 1024 > $OUT/phase_c5.txt 2>&1; tail -n 1 $OUT/phase_c5.txt
