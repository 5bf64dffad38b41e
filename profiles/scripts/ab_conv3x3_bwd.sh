for v in 1 0 1 0 1 0; do
  RSDET_CONV3X3_BWD_AS_FWD=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernels 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bwd-as-fwd $v fp32', round(d['ms_per_step'],2), 'bf16', round(d['bf16']['ms_per_step'],2))"
done
