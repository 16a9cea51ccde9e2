#!/bin/bash
# GPU box: engine-related GPU tests on the product build, then per-phase cycles + bench line for the product build and every
# developer variant libhilo_hip_<tag>.so present (tools: _build.build(tag=...)).   gpurun -- bash tools/gpu/run_c2.sh <tag> [tests]
TAG=${1:-r3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$2" != "notests" ]; then
timeout 900 python -m pytest tests/test_nmpc_gpu.py tests/test_gen_gpu.py tests/test_mhe_gpu.py tests/test_hybrid_gpu.py tests/test_coll_gpu.py tests/test_jit_gpu.py tests/test_tv_gpu.py tests/test_dae_gpu.py tests/test_smpc_gpu.py tests/test_zz_late_gpu.py -m gpu -q --maxfail=12 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -n 25 $OUT/pytest.log
fi
for lib in hilo_mpc_amd/libhilo_hip.so hilo_mpc_amd/libhilo_hip_*.so; do
  [ -f "$lib" ] || continue
  name=$(basename $lib .so)
  echo "== $name"
  if [[ "$name" == "libhilo_hip" ]]; then timeout 300 python tools/phase_profile_c5.py 1024 > $OUT/phase_c5.txt 2>&1; tail -n 1 $OUT/phase_c5.txt; fi
  if [[ "$name" == *dprof* ]]; then
    HILO_LIB_PATH=$PWD/$lib timeout 120 python tools/dbg/dprof.py > $OUT/dprof_$name.txt 2>&1; tail -n 3 $OUT/dprof_$name.txt
  else
    HILO_LIB_PATH=$PWD/$lib timeout 120 python tools/phase_profile.py 4 > $OUT/phase_$name.txt 2>&1; tail -n 1 $OUT/phase_$name.txt
    HILO_LIB_PATH=$PWD/$lib timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; cut -c1-330 $OUT/bench_$name.json
  fi
done
