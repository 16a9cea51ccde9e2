#!/bin/bash
# developer aid: derivative-phase cycles of C5-DAE with parts of a Taylor task compiled out (numbers are wrong then - timing only)
for k in NONE SKIP_ROWS SKIP_QUAD SKIP_LUSOLVE; do
  mkdir -p /tmp/ck_$k
  if [ $k = NONE ]; then opts=""; else opts="-DHILO_DBG_$k"; fi
  echo "== $k"
  HILO_JIT_CACHE=/tmp/ck_$k HILO_JIT_EXTRA_OPTS="$opts" C5DAE=1 HILO_DBG_ONE=1 timeout 600 python tools/phase_profile_c5.py 64 2>&1 | tail -1 | cut -c1-400
done
