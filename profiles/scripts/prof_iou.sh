#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ $v = default ]; then unset RSDET_LIB_PATH; else export RSDET_LIB_PATH=$GRAFT_REPO_ROOT/scratch/lib_$v.so; fi
  rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_iou_$v -o p -- python3 $GRAFT_REPO_ROOT/scratch/iou_time.py $KS > /dev/null 2>&1
  echo "== $v"; python3 $GRAFT_REPO_ROOT/scratch/ktrace_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof_iou_$v
done
