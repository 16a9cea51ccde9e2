#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline numbers (run on the GPU box through gpurun), per configuration:
#   1. kernel trace + stats of `bench.py --config C`              -> gpurun_out/prof_$TAG/$C/trace
#   2. PMC pass FETCH_SIZE (own run, kernel-trace only)           -> gpurun_out/prof_$TAG/$C/pmc_fetch
#   3. PMC pass WRITE_SIZE (own run)                              -> gpurun_out/prof_$TAG/$C/pmc_write
#   4. the bench line itself (un-profiled)                        -> gpurun_out/prof_$TAG/$C/bench_n1.json
# then condense with:  python profiles/summarize.py gpurun_out/prof_$TAG $TAG
# usage: profiles/run_profile.sh r02 "C2 C3-ekf ..." [steps]
TAG=${1:-r02}
CONFIGS=${2:-"C2"}
STEPS=${3:-20}
ROOT=$(pwd)
export TMPDIR=/tmp
for C in $CONFIGS; do
  OUT=$ROOT/gpurun_out/prof_$TAG/$C
  mkdir -p $OUT
  cd /tmp
  BENCH="python $ROOT/bench.py --config $C --no-cpu-baseline"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- $BENCH --steps $STEPS --warmup 5 > $OUT/bench_under_rocprof.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $BENCH --steps 5 --warmup 2 > $OUT/pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $BENCH --steps 5 --warmup 2 > $OUT/pmc_write.log 2>&1
  cd $ROOT
  python bench.py --config $C --steps $STEPS --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
  # keep the merge-back small: only the csv files the summary reads
  find $OUT -name "*agent_info*" -delete
  tail -c 400 $OUT/bench_n1.json
  echo
done
