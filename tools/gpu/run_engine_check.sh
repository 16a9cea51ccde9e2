#!/bin/bash
# After a change to the interior-point engine: sub-phase clocks (needs the -DHILO_OCP_DPROF library), phase cycles, instruction-cache
# counters, the zoo parity tests and the C2 bench line - one gpurun call (~2.5 GPU-minutes).
OUT=gpurun_out/engine_check
mkdir -p $OUT
export TMPDIR=/tmp
HILO_LIB_PATH=$PWD/hilo_mpc_amd/libhilo_hip_dprof.so timeout 120 python tools/dbg/dprof.py > $OUT/dprof.txt 2>&1; tail -n 7 $OUT/dprof.txt
timeout 120 python tools/phase_profile.py 4 > $OUT/phase.txt 2>&1; tail -n 2 $OUT/phase.txt
timeout 600 python -m pytest tests/test_nmpc_gpu.py tests/test_mhe_gpu.py -m gpu -x -q > $OUT/tests.txt 2>&1; tail -n 5 $OUT/tests.txt
python bench.py --steps 20 --warmup 5 > $OUT/C2_20.json 2>/dev/null; cat $OUT/C2_20.json | cut -c1-400
python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/C2_50.json 2>/dev/null; cat $OUT/C2_50.json | cut -c1-300
bash profiles/run_pmc_icache.sh chk > $OUT/icache.log 2>&1
