#!/bin/bash
# GPU box, round 5 second run: smoke, the engine's parity tests, phase / sub-phase cycles, timing experiments of the Riccati stage.
TAG=${1:-r5b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 30 $OUT/smoke.log; exit 1; }
tail -n 1 $OUT/smoke.log
timeout 120 python tools/phase_profile.py 4 > $OUT/phase.txt 2>&1; tail -n 1 $OUT/phase.txt
HILO_LIB_PATH=$PWD/hilo_mpc_amd/libhilo_hip_dprof.so timeout 120 python tools/dbg/dprof.py > $OUT/dprof.txt 2>&1; tail -n 2 $OUT/dprof.txt
for n in 0 1 2 3 4 5 6; do
  [ -f hilo_mpc_amd/libhilo_hip_exp$n.so ] && HILO_LIB_PATH=$PWD/hilo_mpc_amd/libhilo_hip_exp$n.so timeout 120 python tools/dbg/exp_ric.py 2>&1 | tail -n 1 | tee -a $OUT/exp_ric.txt
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_C2.json 2> $OUT/bench_C2.err; cut -c1-330 $OUT/bench_C2.json
timeout 900 python -m pytest tests/test_gen_gpu.py tests/test_coll_gpu.py tests/test_mhe_gpu.py tests/test_lmpc_gpu.py tests/test_nmpc_gpu.py tests/test_dae_gpu.py tests/test_hybrid_gpu.py tests/test_tv_gpu.py -m gpu -q --maxfail=25 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|rc=" $OUT/pytest.log | tail -n 30
