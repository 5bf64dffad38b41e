import csv, sys, glob, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "rsdet" not in name: continue
    acc[(name, r.get("Grid_Size") or r.get("Grid_Size_X"), r.get("Workgroup_Size") or r.get("Workgroup_Size_X"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(acc.items(), key=lambda kv: (kv[0][0], int(kv[0][1]))):
    v = sorted(v)
    print("%-40s grid=%-9s wg=%-4s n=%-4d med=%.1f us min=%.1f" % (k[0][-40:], k[1], k[2], len(v), v[len(v)//2]/1e3, v[0]/1e3))
