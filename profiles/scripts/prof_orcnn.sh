#!/bin/bash
# rocprofv3 kernel stats of the Oriented R-CNN / VAN-B3 step; summary -> gpurun_out/<tag>_orcnn_breakdown.txt
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
rm -rf $O/prof_orcnn
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_orcnn -o p -- python3 $R/bench.py --model orcnn_van3 --no-cpu-baseline --no-kernels --steps 8 --warmup 3 > $O/bench_orcnn_under_rocprof.json 2>/dev/null
python3 $R/profiles/scripts/step_breakdown.py $O/prof_orcnn 0 60 > $O/${TAG}_orcnn_breakdown.txt
cp $(find $O/prof_orcnn -name "*kernel_stats.csv" | head -1) $O/${TAG}_orcnn_kernel_stats.csv
find $O/prof_orcnn -name "*kernel_trace.csv" -delete
tail -1 $O/bench_orcnn_under_rocprof.json | cut -c1-200
cat $O/${TAG}_orcnn_breakdown.txt
