#!/bin/bash
# A/B of the bf16 line: AlignConv in bf16, bias+ReLU fused under autocast
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
run() { tag=$1; shift; env "$@" python3 $R/bench.py --dtype bf16 --no-cpu-baseline --no-kernels --steps 30 --warmup 5 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'])"; }
run base A=1
run acbf16 RSDET_ALIGNCONV_BF16=1
run biasact RSDET_FUSED_BIAS_RELU_AMP=1
run both RSDET_ALIGNCONV_BF16=1 RSDET_FUSED_BIAS_RELU_AMP=1
run base2 A=1
