#!/bin/bash
# default bench line, then rocprofv3 kernel stats of (a) the train step alone and (b) the kernel micro-benches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -1 $O/bench_default.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_step -o p -- python3 $R/bench.py --no-cpu-baseline --no-kernels > $O/bench_step_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_full -o p -- python3 $R/bench.py --no-cpu-baseline > $O/bench_full_under_rocprof.json 2>/dev/null
for d in prof_step prof_full; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); echo "== $d"; test -n "$f" && head -14 "$f" | cut -c1-150; done
f=$(find $O/prof_full -name "*kernel_stats.csv" | head -1); test -n "$f" && grep rsdet "$f" | cut -c1-170
