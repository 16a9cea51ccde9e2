#!/bin/bash
# GPU box: tracking-policy tests, C2 phase profile and bench lines (C2 default run, C4, C3-mhe).   gpurun -- bash tools/gpu/run_c2.sh <tag>
TAG=${1:-c2}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_nmpc_gpu.py tests/test_hybrid_gpu.py tests/test_tv_gpu.py tests/test_jit_gpu.py -m gpu -q --maxfail=20 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|rc=" $OUT/pytest.log | tail -n 10
timeout 120 python tools/phase_profile.py 4 > $OUT/phase.txt 2>&1; tail -n 1 $OUT/phase.txt
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_C2_default.json 2> /dev/null; cut -c1-300 $OUT/bench_C2_default.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_C2.json 2> /dev/null; cut -c1-300 $OUT/bench_C2.json
timeout 300 python bench.py --config C4 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_C4.json 2> /dev/null; cut -c1-300 $OUT/bench_C4.json
