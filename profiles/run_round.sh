#!/bin/bash
# Everything the round's committed evidence comes from, in one call on the GPU box:
#   profiles/run_round.sh r05        then (here)  python profiles/summarize.py gpurun_out/prof_r05 r05 ; python profiles/summarize_pmc.py gpurun_out/pmc_r05 r05
#                                                 python profiles/summarize_pmc.py gpurun_out/pmc_r05_<config> r05 <config>   (C4, C3-mhe, C5, C5-dae, icache)
TAG=${1:-r06}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
profiles/run_profile.sh $TAG "C2 C1 C3-mhe C3-ekf C3-ukf C4 gp-predict" 20
profiles/run_profile.sh $TAG "C5 C5-dae" 8
# issue / matrix-core counters: the headline kernel (+ the GP prediction kernel), then the kernels furthest below their roof
profiles/run_pmc_valu.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1
profiles/run_pmc_valu.sh $TAG C4 4 6 >> gpurun_out/pmc_$TAG.log 2>&1
profiles/run_pmc_valu.sh $TAG C3-mhe 4 6 >> gpurun_out/pmc_$TAG.log 2>&1
profiles/run_pmc_valu.sh $TAG C5 3 3 >> gpurun_out/pmc_$TAG.log 2>&1
profiles/run_pmc_valu.sh $TAG C5-dae 2 2 >> gpurun_out/pmc_$TAG.log 2>&1
profiles/run_pmc_icache.sh $TAG >> gpurun_out/pmc_$TAG.log 2>&1      # instruction-cache hit rate of the headline kernel (87 KB of code)
# the driver's command (20 steps) and a longer timed region (50 steps) of the headline configuration, and the batch sweep
mkdir -p gpurun_out/bench_$TAG
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG/C2_20.json 2>/dev/null
python bench.py --steps 50 --warmup 5 > gpurun_out/bench_$TAG/C2_50.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --batch 16384 --no-cpu-baseline > gpurun_out/bench_$TAG/C2_B16384.json 2>/dev/null
python bench.py --config C3-ekf --batch 1048576 --no-cpu-baseline > gpurun_out/bench_$TAG/C3-ekf_B1M.json 2>/dev/null
python bench.py --config C3-ukf --batch 1048576 --no-cpu-baseline > gpurun_out/bench_$TAG/C3-ukf_B1M.json 2>/dev/null
python tools/phase_profile.py 4 > gpurun_out/bench_$TAG/phase_cycles.txt 2>&1 || true
python tools/phase_profile.py 3 C4 >> gpurun_out/bench_$TAG/phase_cycles.txt 2>&1 || true
# sub-phase clocks: needs the developer variant of the library, built here (no GPU needed) with
#   python -c "from hilo_mpc_amd import _build; _build.build(tag='dprof', extra_flags=['-DHILO_OCP_DPROF'])"
# and removed again after the evidence run (it is not part of the product)
[ -f hilo_mpc_amd/libhilo_hip_dprof.so ] && HILO_LIB_PATH=$PWD/hilo_mpc_amd/libhilo_hip_dprof.so python tools/dbg/dprof.py >> gpurun_out/bench_$TAG/phase_cycles.txt 2>&1
ls gpurun_out/prof_$TAG gpurun_out/bench_$TAG
