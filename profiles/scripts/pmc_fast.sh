#!/bin/bash
# SQ counters + kernel trace of iou_fast_tile_kernel: pmc_fast.sh <tag>   (RSDET_LIB_PATH selects the build)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_fast_$1
rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -o p -- python3 $R/profiles/scripts/iou_fast_step.py > $O/g$i.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- python3 $R/profiles/scripts/iou_fast_step.py > $O/trace.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, os, collections, sys
O=sys.argv[1]
acc=collections.defaultdict(list)
for f in glob.glob(O+"/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "iou_fast" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
c={k: sum(v)/len(v) for k,v in acc.items()}
dur=[]
for f in glob.glob(O+"/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "iou_fast" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
dur.sort()
out=open(O+"/summary.txt","w")
def p(*a):
    s=" ".join(str(x) for x in a); print(s); out.write(s+"\n")
p("iou_fast_tile_kernel: n=%d median %.2f us  min %.2f  mean %.2f" % (len(dur), dur[len(dur)//2], dur[0], sum(dur)/len(dur)))
for k in sorted(c): p("  %-24s %.4g" % (k, c[k]))
w=c.get("SQ_WAVES",1); wc=c.get("SQ_WAVE_CYCLES",1)
p("  VALU per wave %.0f | VALU active %.1f%% of wave-cycles | waiting %.1f%% | issue stall %.1f%%" % (c.get("SQ_INSTS_VALU",0)/w, 100*c.get("SQ_ACTIVE_INST_VALU",0)/wc, 100*c.get("SQ_WAIT_ANY",0)/wc, 100*c.get("SQ_WAIT_INST_ANY",0)/wc))
PY
