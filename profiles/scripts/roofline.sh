#!/bin/bash
# One kernel-trace pass + the two PMC passes (WRITE_SIZE, FETCH_SIZE: separate passes, --kernel-trace only beside them,
# MI355X_MICROARCH.md) over profiles/scripts/pmc_kernels.py, then profiles/scripts/roofline.py -> <tag>_roofline.json.
# On the GPU box:  gpurun -- 'bash profiles/scripts/roofline.sh r02'   (outputs under gpurun_out/roofline/)
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/roofline
rm -rf $O; mkdir -p $O
export RSDET_ROOFLINE_DIR=$O
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o p -- python3 $R/profiles/scripts/pmc_kernels.py > $O/trace.log 2>&1
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 $R/profiles/scripts/pmc_kernels.py > $O/pmc_$c.log 2>&1
done
# SQ instruction / wait counters of the IoU kernels (the issue-bound evidence VERDICT r2 item 4 asks for): two passes
# of 8 SQ counters each over the same driver
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/sq$i -o p -- python3 $R/profiles/scripts/pmc_kernels.py > $O/sq$i.log 2>&1
done
python3 $R/profiles/scripts/roofline.py $O $O/${TAG}_roofline.json
python3 $R/profiles/scripts/sq_table.py $O $O/${TAG}_iou_sq_counters.txt
