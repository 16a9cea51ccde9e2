#!/bin/bash
# Filters in the large-batch regime (B = 2^20 and 2^16, 16 steps per launch): the product library, optionally against developer variants
# (python -c "from hilo_mpc_amd import _build; _build.build(tag='<tag>', extra_flags=[...])"; tags as arguments), the general
# kernel (HILO_KF_LEAN=0), then the issue counters of the multi-step kernels (own --pmc passes).
OUT=gpurun_out/kf_b1m
mkdir -p $OUT
for k in ekf ukf; do
  python bench.py --config C3-$k --batch 1048576 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/$k.json
  python bench.py --config C3-$k --batch 65536 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${k}_64k.json
  HILO_KF_LEAN=0 python bench.py --config C3-$k --batch 1048576 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${k}_general.json
  for t in "$@"; do
    HILO_LIB_PATH=$(pwd)/hilo_mpc_amd/libhilo_hip_$t.so python bench.py --config C3-$k --batch 1048576 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${k}_$t.json
  done
done
python - "$@" <<'PY'
import json, sys
for k in ('ekf', 'ukf'):
    for t in ['', '_64k', '_general'] + ['_' + a for a in sys.argv[1:]]:
        d = json.load(open(f'gpurun_out/kf_b1m/{k}{t}.json'))
        print(k + t, f"{d['value']:.4g}", d['roofline']['kernel_ms'])
PY
for k in ekf ukf; do
  BENCH_ARGS="--batch 1048576" OUT_SUFFIX=_B1M bash profiles/run_pmc_valu.sh r06 C3-$k 4 2 > $OUT/pmc_$k.log 2>&1
done
