#!/bin/bash
# SQ counters of the IoU kernels (one rocprofv3 --pmc pass per counter group; --kernel-trace only beside it).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_iou
rm -rf $O; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -o p -- python3 $R/profiles/scripts/iou_step_shape.py "$@" > $O/g$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_iou"
for g in sorted(glob.glob(O+"/g*/")):
    files=glob.glob(g+"/**/*counter_collection.csv", recursive=True)
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        if not any(s in k for s in ("iou","assign","at_")): continue
        print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, "n=%d"%len(next(iter(v.values()))))
PY
