#!/bin/bash
TAG=${1:-qp}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lmpc_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|rc=|Error" $OUT/pytest.log | tail -n 20
timeout 300 python tools/dbg/qp_prof.py 2>&1 | grep -v amdgpu.ids | tee $OUT/qp_prof.txt
timeout 600 python tools/dbg/qp_time.py 2>&1 | grep -v amdgpu.ids | tee $OUT/qp_time.txt
timeout 600 python bench.py --config C1 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_C1.json 2> $OUT/bench_C1.err; cut -c1-900 $OUT/bench_C1.json; tail -n 3 $OUT/bench_C1.err
