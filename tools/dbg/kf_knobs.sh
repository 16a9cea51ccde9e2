#!/bin/bash
# Debug aid: rocprofv3 duration of the team EKF kernel with parts switched off (developer builds libhilo_hip_kfd<bits>.so,
# csrc/hilo_kf_kernel.h HILO_KF_DBG: 1 = no model evaluation, 2 = no gain, 4 = no matrix phases)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
KIND=${1:-C3-ekf}
for v in 0 1 2 4 7; do
  d=gpurun_out/kfknob_$v; rm -rf $d
  lib=$PWD/hilo_mpc_amd/libhilo_hip.so; [ $v != 0 ] && lib=$PWD/hilo_mpc_amd/libhilo_hip_kfd$v.so
  HILO_LIB_PATH=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python bench.py --config $KIND --no-cpu-baseline --steps 30 > $d.log 2>&1
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  python - "$f" $v <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'kf_' in r['Name']:
        print('dbg', sys.argv[2], r['Name'][:50], r['Calls'], 'avg ns', r['AverageNs'], 'min', r['MinNs'])
PY
done
