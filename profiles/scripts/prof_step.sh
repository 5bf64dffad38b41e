#!/bin/bash
# rocprofv3 kernel stats of the default bench (fp32) and the bf16 line; summaries -> gpurun_out/prof_<tag>_{f32,bf16}.txt
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for dt in ${2:-f32 bf16}; do
  rm -rf $O/prof_step_$dt
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_step_$dt -o p -- python3 $R/bench.py --dtype $dt --no-cpu-baseline --no-kernels --no-bf16-leg --no-extra-legs --steps 10 --warmup 3 > $O/${TAG}_bench_under_rocprof_$dt.json 2>/dev/null
  python3 $R/profiles/scripts/step_breakdown.py $O/prof_step_$dt 29 > $O/${TAG}_step_breakdown_$dt.txt
  cp $(find $O/prof_step_$dt -name "*kernel_stats.csv" | head -1) $O/${TAG}_train_step_kernel_stats_$dt.csv
  find $O/prof_step_$dt -name "*kernel_trace.csv" -delete     # (26 MB each; the stats file carries what the summaries use)
  echo "== $dt"; python3 -c "import json,sys; d=json.loads([l for l in open('$O/${TAG}_bench_under_rocprof_$dt.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"; cat $O/${TAG}_step_breakdown_$dt.txt
done
