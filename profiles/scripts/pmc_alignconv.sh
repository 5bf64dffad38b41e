#!/bin/bash
# PMC passes over the implicit-GEMM AlignConv (level 0 shape): matrix-core busy cycles, vector instruction counts, LDS
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_acm
rm -rf $O; mkdir -p $O
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$tag -o p -- python3 $R/profiles/scripts/alignconv_mfma_check.py > $O/$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_acm"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "alignconv_fwd_mfma_kernel<unsigned short, true>" in r["Kernel_Name"] and r.get("Grid_Size", "0") not in ("", None):
            acc[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for grid, cs in sorted(acc.items(), key=lambda kv: -int(kv[0]))[:2]:
    print("grid", grid, {k: round(sum(v) / len(v)) for k, v in cs.items()})
PY
