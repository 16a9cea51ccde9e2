#!/bin/bash
# Where the cycles of configuration 5's DAE problem go, without instrumenting the kernel: compile variants that do ONE part twice
# (same numbers, same iterations) and read the phase clocks - the difference to the plain build is that part's cost per iteration.
#   tools/dbg/c5dae_twice.sh [batch]     (on the GPU box; each variant compiles with hiprtc in ~20 s)
B=${1:-1024}
export HILO_JIT_CACHE=/tmp/jc_twice
mkdir -p $HILO_JIT_CACHE
for v in "" "-DHILO_DBG_TWICE_VALS" "-DHILO_DBG_TWICE_COLL" "-DHILO_DBG_TWICE_DIRS"; do
  rm -rf $HILO_JIT_CACHE/*
  echo "variant: ${v:-plain}"
  HILO_JIT_EXTRA_OPTS="$v" C5DAE=1 HILO_DBG_ONE=1 python tools/phase_profile_c5.py $B 2>&1 | grep -v amdgpu.ids
done
