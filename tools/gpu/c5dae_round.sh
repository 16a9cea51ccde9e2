#!/bin/bash
# developer round for configuration 5's DAE problem on the GPU box: parity tests of the collocation policies, phase clocks, the bench
# line and the HBM traffic counters (own --pmc passes)          usage: tools/gpu/c5dae_round.sh <out tag> [skip-tests]
TAG=${1:-x}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
if [ -z "$2" ]; then
  python -m pytest tests/test_c5dae_gpu.py tests/test_coll_gpu.py tests/test_custom_gpu.py tests/test_dae_gpu.py -m gpu -x -q 2>&1 | tail -15 > $O/tests.log
  cat $O/tests.log
fi
C5DAE=1 python tools/phase_profile_c5.py 1024 2>&1 | grep -v amdgpu.ids > $O/phase_1024.txt
cat $O/phase_1024.txt
python bench.py --config C5-dae --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_c5dae.json 2> $O/bench_c5dae.err
python - <<PY
import json
d = json.load(open("$O/bench_c5dae.json"))
print("C5-dae", round(d["value"]), "steps/s", round(d["ms_per_step"], 1), "ms", "iters", d["config"]["mean_ipm_iters"], "ok", d["config"]["frac_status_1_or_2"])
PY
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o x -- python $R/bench.py --config C5-dae --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_$c.log 2>&1
done
cd $R
find $O -name "*agent_info*" -delete
python - <<PY
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$O/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "hilo_user_solve" in r["Kernel_Name"]]
        print(c, "GB per launch", [round(float(r["Counter_Value"]) * 1024 / 1e9, 1) for r in rows], "scratch", rows[0].get("Scratch_Size"), "vgpr", rows[0].get("VGPR_Count"), rows[0].get("Accum_VGPR_Count"))
PY
