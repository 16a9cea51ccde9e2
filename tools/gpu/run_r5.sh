#!/bin/bash
# GPU box, round 5: smoke first (a broken engine must not burn the budget), then the chosen tests, phase / sub-phase cycles, bench lines.
#   gpurun -- bash tools/gpu/run_r5.sh <tag> "<pytest selection or 'tests'>" [bench configs ...]
TAG=${1:-r5}; SEL=${2:-tests}; shift; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 30 $OUT/smoke.log; exit 1; }
tail -n 1 $OUT/smoke.log
timeout 120 python tools/phase_profile.py 4 > $OUT/phase.txt 2>&1; tail -n 1 $OUT/phase.txt
if [ -f hilo_mpc_amd/libhilo_hip_dprof.so ]; then
  HILO_LIB_PATH=$PWD/hilo_mpc_amd/libhilo_hip_dprof.so timeout 120 python tools/dbg/dprof.py > $OUT/dprof.txt 2>&1; tail -n 2 $OUT/dprof.txt
fi
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_C2.json 2> $OUT/bench_C2.err; cut -c1-330 $OUT/bench_C2.json
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_C2_default.json 2> /dev/null; cut -c1-330 $OUT/bench_C2_default.json
timeout 1500 python -m pytest $SEL -m gpu -q --maxfail=25 -p no:cacheprovider --durations=15 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|rc=|Error" $OUT/pytest.log | tail -n 30
grep -A18 "slowest" $OUT/pytest.log | head -n 20
timeout 120 python tools/phase_profile.py 3 C4 > $OUT/phase_c4.txt 2>&1; tail -n 1 $OUT/phase_c4.txt
for cfg in "$@"; do
  timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err; cut -c1-600 $OUT/bench_$cfg.json; tail -n 2 $OUT/bench_$cfg.err
done
