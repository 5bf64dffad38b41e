#!/bin/bash
# Kernel launches per bf16 train step (rocprofv3 --kernel-trace over profiles/scripts/host_bound.py: 5 + 30 steps):
#   bash profiles/scripts/count_launches.sh [bf16params]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/launches_${1:-autocast}
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 profiles/scripts/host_bound.py bf16 $1 > $O/run.log 2>&1
tail -1 $O/run.log
python3 - "$O" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
n = len(rows)
c = collections.Counter(r["Kernel_Name"].split("(")[0][:70] for r in rows)
print("kernel launches: %d in 35 steps + set-up = %.0f per step" % (n, n / 35.0))
for k, v in c.most_common(14):
    print("  %6.1f /step  %s" % (v / 35.0, k))
PY
