#!/bin/bash
# GPU box: full -m gpu suite, then the C5-DAE phase profile and bench lines of the configurations that share the engine
TAG=${1:-r5e}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 30 $OUT/smoke.log; exit 1; }
tail -n 1 $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|rc=" $OUT/pytest.log | tail -n 30
C5DAE=1 timeout 300 python tools/phase_profile_c5.py 1024 > $OUT/phase_c5dae.txt 2>&1; tail -n 1 $OUT/phase_c5dae.txt
timeout 300 python tools/phase_profile_c5.py 1024 > $OUT/phase_c5.txt 2>&1; tail -n 1 $OUT/phase_c5.txt
for cfg in C5-dae C5; do
  timeout 600 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err; cut -c1-400 $OUT/bench_$cfg.json
done
