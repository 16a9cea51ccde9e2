#!/bin/bash
# A/B of hilo_models.h::rk4_classic against the generic Runge-Kutta map (-DHILO_RK4_GENERIC; library built with
# _build.build(tag='rkgen', extra_flags=['-DHILO_RK4_GENERIC']), run-time compiled kernels through HILO_JIT_EXTRA_OPTS and a private cache)
OUT=gpurun_out/rk4_ab
mkdir -p $OUT
run() { python bench.py --config $1 --no-cpu-baseline --steps $2 --warmup $3 2>/dev/null | tail -1; }
for c in "C2 20 5" "C3-mhe 10 3" "C4 10 3" "C5 6 2"; do
  set -- $c
  run $1 $2 $3 > $OUT/$1_new.json
  HILO_LIB_PATH=$(pwd)/hilo_mpc_amd/libhilo_hip_rkgen.so HILO_JIT_EXTRA_OPTS="-DHILO_RK4_GENERIC" HILO_JIT_CACHE=/tmp/jit_rkgen run $1 $2 $3 > $OUT/$1_gen.json
  run $1 $2 $3 > $OUT/$1_new2.json
done
python - <<'PY'
import json
for c in ('C2', 'C3-mhe', 'C4', 'C5'):
    r = []
    for t in ('new', 'gen', 'new2'):
        try:
            d = json.load(open(f'gpurun_out/rk4_ab/{c}_{t}.json')); r.append(f"{t} {d['value']:.5g} ({d['roofline']['kernel_ms']:.4g} ms)")
        except Exception as e:
            r.append(f'{t} failed {e}')
    print(c, ' | '.join(r))
PY
