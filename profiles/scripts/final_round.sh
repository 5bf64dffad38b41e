#!/bin/bash
# end-of-round measurements: the bench lines of every configuration + rocprofv3 kernel stats of the fp32 and bf16 steps
TAG=${1:-r03_k}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python3 bench.py --dtype bf16 --no-cpu-baseline --steps 30 --warmup 5 > $O/${TAG}_bench_bf16.json 2> $O/${TAG}_bench_bf16.err
python3 bench.py --model s2anet_r101 --dtype bf16 --no-cpu-baseline --steps 20 --warmup 5 > $O/${TAG}_bench_r101_bf16.json 2>/dev/null
python3 bench.py --model orcnn_van3 --no-cpu-baseline --steps 10 --warmup 3 > $O/${TAG}_bench_orcnn.json 2>/dev/null
python3 bench.py --gpus 2 --no-cpu-baseline --no-kernels --steps 5 --warmup 2 > $O/${TAG}_bench_gpus2_gloo.json 2>/dev/null
for f in bench bench_bf16 bench_r101_bf16 bench_orcnn bench_gpus2_gloo; do grep '^{' $O/${TAG}_$f.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['n_gpus'], round(d['value'],2), round(d['ms_per_step'],2), (d.get('roofline') or {}).get('frac'))"; done
bash profiles/scripts/prof_step.sh $TAG
# Oriented R-CNN VAN-B3 under rocprofv3: kernel stats + category breakdown (12 steps: 4 set-up + 3 warm-up + 5 timed)
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_orcnn
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_orcnn -o p -- python3 $R/bench.py --model orcnn_van3 --no-cpu-baseline --no-kernels --steps 5 --warmup 3 > $O/${TAG}_bench_orcnn_under_rocprof.json 2>/dev/null
find $O/prof_orcnn -name "*kernel_trace.csv" -delete
python3 $R/profiles/scripts/step_breakdown.py $O/prof_orcnn 12 > $O/${TAG}_orcnn_breakdown.txt
cp $(find $O/prof_orcnn -name "*kernel_stats.csv" | head -1) $O/${TAG}_orcnn_kernel_stats.csv
head -8 $O/${TAG}_orcnn_breakdown.txt
