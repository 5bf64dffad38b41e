#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -o p -- python3 $R/bench.py --dtype bf16 --no-cpu-baseline --no-kernels --steps 10 --warmup 3 > $O/bench_bf16_prof.json 2>/dev/null
tail -1 $O/bench_bf16_prof.json | cut -c1-200
