"""Per-kernel averages of the SQ counter passes of profiles/scripts/roofline.sh (IoU / anchor-target kernels):
instructions issued, busy / wait cycles, and the derived 'VALU instructions per wave', 'share of wave-cycles in which
a VALU instruction was active' and 'share spent waiting' -- the issue-bound evidence for the reference-order clipper.
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import os
import sys


def main(d, out):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "sq*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            if any(s in k for s in ("iou", "at_", "assign")):
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines = []
    for k in sorted(acc):
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        us = sorted(dur.get(k, [0]))[len(dur.get(k, [0])) // 2]
        waves = c.get("SQ_WAVES", 0) or 1
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        lines.append("%s  (median %.1f us)" % (k, us))
        lines.append("  waves %.0f | VALU insts %.3g (%.0f per wave) | LDS insts %.3g | SALU %.3g | VMEM %.3g | SMEM %.3g" % (
            waves, c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_VALU", 0) / waves, c.get("SQ_INSTS_LDS", 0),
            c.get("SQ_INSTS_SALU", 0), c.get("SQ_INSTS_VMEM", 0), c.get("SQ_INSTS_SMEM", 0)))
        lines.append("  wave-cycles %.3g | VALU active %.1f %% | any inst active %.1f %% | waiting (s_waitcnt / barrier) %.1f %% | "
                     "issue stall %.1f %% | LDS bank conflict cycles %.3g" % (
                         wc, 100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                         100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc,
                         c.get("SQ_LDS_BANK_CONFLICT", 0)))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
