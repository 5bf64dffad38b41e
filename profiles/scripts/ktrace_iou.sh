#!/bin/bash
# per-kernel durations of the IoU / anchor-target forms at the step shape (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ktrace_iou
rm -rf $O; mkdir -p $O
REPS=20 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $R/profiles/scripts/iou_step_shape.py "$@" > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/ktrace_iou"
f=glob.glob(O+"/**/*kernel_trace.csv", recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][:70]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(acc.items(), key=lambda kv:-sum(kv[1])):
    v=sorted(v); print("%-72s n=%3d  median %.2f us  min %.2f" % (k, len(v), v[len(v)//2], v[0]))
PY
