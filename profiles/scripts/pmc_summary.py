"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files (values are KB per launch for *_SIZE)."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "rsdet" in name:
                acc[(r["Counter_Name"], name)].append(float(r["Counter_Value"]))
        for (c, k), v in sorted(acc.items()):
            print("%-11s %-48s launches=%d  avg_KB=%.0f  (%.1f MB)" % (c, k[-48:], len(v), sum(v) / len(v), sum(v) / len(v) / 1024))
