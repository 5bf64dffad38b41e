DB=$PWD/gpurun_out/miopen_db_cl
mkdir -p $DB; cp rs_detection_amd/miopen_db/*.txt $DB/
export MIOPEN_USER_DB_PATH=$DB
s=$(date +%s)
RSDET_CUDNN_BENCHMARK=1 MIOPEN_FIND_MODE=NORMAL timeout 1500 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernels --no-bf16-leg --memory-format channels_last 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('search run fp32 channels_last', round(d['ms_per_step'],2))"
echo "search wall $(( $(date +%s) - s )) s"
for mf in channels_last contiguous; do
python bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-kernels --no-bf16-leg --memory-format $mf 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32 $mf', round(d['ms_per_step'],2))"
done
