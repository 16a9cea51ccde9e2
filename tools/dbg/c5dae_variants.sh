#!/bin/bash
# bench line of configuration 5's DAE problem for build variants of its run-time compiled kernel (HILO_JIT_EXTRA_OPTS is not part of
# the cache key: one private cache directory per variant)        tools/dbg/c5dae_variants.sh "<opts 1>" "<opts 2>" ...
i=0
for v in "$@"; do
  i=$((i+1))
  export HILO_JIT_CACHE=/tmp/jc_var_$i
  rm -rf $HILO_JIT_CACHE; mkdir -p $HILO_JIT_CACHE
  echo "variant: $v"
  HILO_JIT_EXTRA_OPTS="$v" python bench.py --config C5-dae --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['value']),'steps/s',round(d['ms_per_step'],1),'ms iters',round(d['config']['mean_ipm_iters'],2),'ok',d['config']['frac_status_1_or_2'])"
done
