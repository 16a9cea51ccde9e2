#!/bin/bash
for cfg in "64 1" "128 2" "128 1"; do
  set -- $cfg
  HILO_EXTRA_FLAGS="-DHILO_OCP_TPB=$1 -DHILO_OCP_MINW=$2" python -m hilo_mpc_amd._build --force > /dev/null 2>&1
  echo "== TPB=$1 MINW=$2"
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('steps/s %.3e  kernel_ms %.3f  iters %.2f ok %.3f'%(d['value'], d['roofline']['kernel_ms'], d['config']['mean_ipm_iters'], d['config']['frac_status_1_or_2']))"
  python gpurun_phase.py | tail -1
done
