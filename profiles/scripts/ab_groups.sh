for mode in split all; do
  for rep in 1 2; do
  RSDET_S2A_GROUPS=$mode python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode fp32', round(d['ms_per_step'],2), 'bf16 leg', round(d['bf16']['ms_per_step'],2))"
  done
done
RSDET_S2A_PACKED=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernels 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('loop fp32', round(d['ms_per_step'],2), 'bf16 leg', round(d['bf16']['ms_per_step'],2))"
