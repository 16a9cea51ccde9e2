#!/bin/bash
# GPU box: full -m gpu suite, per-phase cycle profile of the C2 solve, default bench line.  Usage: gpurun -- bash tools/gpu/run_all.sh <tag>
TAG=${1:-r3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -n 40 $OUT/pytest.log
timeout 120 python tools/phase_profile.py 4 > $OUT/phase.txt 2>&1; cat $OUT/phase.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_c2.json 2> $OUT/bench_c2.err; cat $OUT/bench_c2.json | cut -c1-900
