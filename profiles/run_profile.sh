#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline numbers (run on the GPU box through gpurun):
#   1. kernel trace + stats of the default bench command        -> gpurun_out/prof_$TAG/trace
#   2. PMC pass FETCH_SIZE (own run, kernel-trace only)           -> gpurun_out/prof_$TAG/pmc_fetch
#   3. PMC pass WRITE_SIZE (own run)                              -> gpurun_out/prof_$TAG/pmc_write
# then condense with:  python profiles/summarize.py gpurun_out/prof_$TAG $TAG <timed launches>
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $TAG -- $BENCH > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- $BENCH --steps 5 --warmup 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- $BENCH --steps 5 --warmup 2 > $OUT/pmc_write.log 2>&1
cd $ROOT
find $OUT -name "*.csv" | head -20
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -c 600 $OUT/bench_n1.json
