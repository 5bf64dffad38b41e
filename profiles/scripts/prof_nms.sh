#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_nms -o p -- python3 $GRAFT_REPO_ROOT/scratch/nms_time.py $KS > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/scratch/ktrace_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof_nms
