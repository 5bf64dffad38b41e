#!/bin/bash
# Extends the packaged MIOpen find records (rs_detection_amd/miopen_db) with the shapes of the current step:
# seeds a scratch user database with the packaged files, lets MIOpen search (benchmark mode + FIND_MODE=NORMAL) through
# the fp32 step and the bf16 leg of bench.py, and leaves the grown files in gpurun_out/miopen_db for copying back.
#   gpurun --timeout 2400 -- 'bash profiles/scripts/tune_miopen.sh'
set -u
DB=$PWD/gpurun_out/miopen_db
mkdir -p $DB
cp rs_detection_amd/miopen_db/*.txt $DB/
export MIOPEN_USER_DB_PATH=$DB RSDET_CUDNN_BENCHMARK=1 MIOPEN_FIND_MODE=NORMAL
s=$(date +%s)
timeout 1500 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernels --memory-format channels_last 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32', round(d['ms_per_step'],2), 'bf16 leg', round(d['bf16']['ms_per_step'],2))"
echo "search wall $(( $(date +%s) - s )) s"
for extra in "$@"; do
  timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernels $extra 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$extra', round(d['ms_per_step'],2))"
done
ls -la $DB
unset RSDET_CUDNN_BENCHMARK MIOPEN_FIND_MODE
python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-kernels --memory-format channels_last 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('with the new records: fp32', round(d['ms_per_step'],2), 'bf16 leg', round(d['bf16']['ms_per_step'],2))"
