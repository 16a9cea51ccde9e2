#!/bin/bash
# Instruction-issue and matrix-core counters of a configuration's dominant kernel (own rocprofv3 passes, kernel-trace only):
#   [BENCH_ARGS="--batch 1048576" OUT_SUFFIX=_B1M] bash profiles/run_pmc_valu.sh <tag> [config [steps warmup]]      then      python profiles/summarize_pmc.py gpurun_out/pmc_<tag>[_<config>] <tag> [config]
# config C2 (default) also runs the matrix-core pass of the GP prediction kernel (the other place MFMA is used).
TAG=${1:-r01}
CFG=${2:-C2}
STEPS=${3:-6}
WARM=${4:-14}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
[ "$CFG" != "C2" ] && OUT=${OUT}_$CFG
OUT=${OUT}${OUT_SUFFIX:-}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --config $CFG --no-cpu-baseline --steps $STEPS --warmup $WARM ${BENCH_ARGS:-}"
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$name -o $TAG -- $BENCH > $OUT/$name.log 2>&1 || echo "failed: $set"
done
if [ "$CFG" = "C2" ]; then
  # the matrix-core counters of the GP prediction kernel, own pass
  GP="python $ROOT/bench.py --config gp-predict --no-cpu-baseline --steps 6 --warmup 4"
  mkdir -p $OUT/gp
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/gp/mfma -o $TAG -- $GP > $OUT/gp/mfma.log 2>&1 || echo "failed: gp mfma"
fi
cd $ROOT
find $OUT -name "*agent_info*" -delete
find $OUT -name "*counter_collection.csv" | head
