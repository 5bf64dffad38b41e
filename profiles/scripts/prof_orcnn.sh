#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_orcnn -o p -- python3 $R/bench.py --model orcnn_van3 --no-cpu-baseline --no-kernels --steps 5 --warmup 3 > $O/bench_orcnn.json 2>/dev/null
tail -1 $O/bench_orcnn.json | cut -c1-200
