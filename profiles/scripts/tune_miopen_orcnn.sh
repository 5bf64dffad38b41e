#!/bin/bash
# Extends the packaged MIOpen find records with the shapes of the Oriented R-CNN VAN-B3 step (fp32 NCHW): seeds a
# scratch user database with the packaged files, lets MIOpen search (benchmark mode + FIND_MODE=NORMAL), and leaves
# the grown files in gpurun_out/miopen_db_orcnn for copying back.
#   gpurun --timeout 3000 -- 'bash profiles/scripts/tune_miopen_orcnn.sh'
set -u
DB=$PWD/gpurun_out/miopen_db_orcnn
mkdir -p $DB
cp rs_detection_amd/miopen_db/*.txt $DB/
python bench.py --model orcnn_van3 --steps 10 --warmup 3 --no-cpu-baseline --no-kernels 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('before', round(d['ms_per_step'],2))"
export MIOPEN_USER_DB_PATH=$DB RSDET_CUDNN_BENCHMARK=1 MIOPEN_FIND_MODE=NORMAL
s=$(date +%s)
timeout ${1:-2000} python bench.py --model orcnn_van3 --steps 2 --warmup 1 --no-cpu-baseline --no-kernels 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('searching run', round(d['ms_per_step'],2))"
echo "search wall $(( $(date +%s) - s )) s"
ls -la $DB
unset RSDET_CUDNN_BENCHMARK MIOPEN_FIND_MODE
for i in 1 2; do python bench.py --model orcnn_van3 --steps 10 --warmup 3 --no-cpu-baseline --no-kernels 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('with the new records', round(d['ms_per_step'],2))"; done
