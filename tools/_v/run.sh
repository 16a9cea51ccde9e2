for v in base inl; do
  cp tools/_v/lib_$v.so hilo_mpc_amd/libhilo_hip.so
  echo "== $v"; python tools/phase_profile.py 2>&1 | tail -1
  python bench.py --steps 50 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])"
done
cp tools/_v/lib_inl.so hilo_mpc_amd/libhilo_hip.so; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
