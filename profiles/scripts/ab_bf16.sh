#!/bin/bash
# A/B of the bf16 line inside ONE box (boxes differ by 1-3 %): alternate the two settings
# usage: ab_bf16.sh VAR   -> runs VAR=1 / VAR=0 alternately, 3 times each
R=$GRAFT_REPO_ROOT
V=${1:-RSDET_CONV2D_BIAS}
for i in 1 2 3; do for v in 1 0; do
  env $V=$v python3 $R/bench.py --dtype bf16 --no-cpu-baseline --no-kernels --steps 30 --warmup 5 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V=$v', round(d['ms_per_step'],3))"
done; done
