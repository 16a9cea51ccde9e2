#!/bin/bash
# Instruction-cache counters of the headline kernel (own rocprofv3 passes, kernel-trace only):
#   bash profiles/run_pmc_icache.sh <tag>     then     python profiles/summarize_pmc.py gpurun_out/pmc_<tag>_icache <tag> icache
TAG=${1:-r05}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_${TAG}_icache
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --config C2 --no-cpu-baseline --steps 6 --warmup 14"
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$name -o $TAG -- $BENCH > $OUT/$name.log 2>&1 || echo "failed: $set"
done
cd $ROOT
find $OUT -name "*agent_info*" -delete
find $OUT -name "*counter_collection.csv" | head
