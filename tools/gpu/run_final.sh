#!/bin/bash
# GPU box, end of a round: full -m gpu suite, rocprofv3 evidence per configuration (profiles/run_profile.sh), phase profiles,
# the batch sweep of C2 and the bandwidth-bound filter runs.   gpurun -- bash tools/gpu/run_final.sh <tag>
TAG=${1:-r03}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 -p no:cacheprovider -s > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|ERROR|hiprtc seconds|rc=" $OUT/pytest.log | tail -n 20
bash profiles/run_profile.sh $TAG "C2 C5 C4 C3-mhe C3-ekf C3-ukf C1 gp-predict" 20 > $OUT/run_profile.log 2>&1
tail -n 40 $OUT/run_profile.log | cut -c1-300
timeout 120 python tools/phase_profile.py 4 > $OUT/phase.txt 2>&1; tail -n 1 $OUT/phase.txt
timeout 300 python tools/phase_profile_c5.py 1024 > $OUT/phase_c5.txt 2>&1; tail -n 1 $OUT/phase_c5.txt
timeout 120 python tools/phase_profile.py 3 C4 > $OUT/phase_c4.txt 2>&1; tail -n 1 $OUT/phase_c4.txt
for b in 2048 4096 8192 16384; do
  timeout 300 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_C2_B$b.json 2> /dev/null; cut -c1-200 $OUT/bench_C2_B$b.json
done
for k in ekf ukf; do
  timeout 300 python bench.py --config C3-$k --batch 1048576 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_C3-${k}_B1M.json 2> /dev/null; cut -c1-1200 $OUT/bench_C3-${k}_B1M.json
done
