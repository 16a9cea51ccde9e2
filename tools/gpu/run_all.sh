#!/bin/bash
# GPU box: full -m gpu suite, then per-phase cycle profiles (C2, C5, C4), sub-phase timers of the debug variant if present, bench lines.
#   gpurun -- bash tools/gpu/run_all.sh <tag> [extra bench configs...]
TAG=${1:-r3}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -s > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|hiprtc seconds|rc=" $OUT/pytest.log | tail -n 30
timeout 120 python tools/phase_profile.py 4 > $OUT/phase.txt 2>&1; tail -n 1 $OUT/phase.txt
timeout 300 python tools/phase_profile_c5.py 1024 > $OUT/phase_c5.txt 2>&1; tail -n 1 $OUT/phase_c5.txt
timeout 120 python tools/phase_profile.py 3 C4 > $OUT/phase_c4.txt 2>&1; tail -n 1 $OUT/phase_c4.txt
if [ -f hilo_mpc_amd/libhilo_hip_dprof.so ]; then
  HILO_LIB_PATH=$PWD/hilo_mpc_amd/libhilo_hip_dprof.so timeout 120 python tools/dbg/dprof.py > $OUT/dprof.txt 2>&1; tail -n 2 $OUT/dprof.txt
fi
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_C2.json 2> $OUT/bench_C2.err; cut -c1-400 $OUT/bench_C2.json
for cfg in "$@"; do
  timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err; cut -c1-600 $OUT/bench_$cfg.json
done
