"""the current kernels-against-bounds table, generated.

    python3 profiles/scripts/bounds_table.py [tag]        (default tag r06)

reads   profiles/<tag>_bench_kernels.json   bench.py's per-kernel table (HIP events around a replayed graph, same process as
                                            the timed step; `kernels_file` of the bench line)
        profiles/<tag>_roofline.json        rocprofv3 --kernel-trace averages + --pmc WRITE_SIZE / FETCH_SIZE traffic of the
                                            same launches (profiles/scripts/roofline.sh), when present
writes  profiles/<tag>_bounds_table.md and splices it into DESIGN.md between the BOUNDS_TABLE markers."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
rows = json.load(open(os.path.join(ROOT, "profiles", tag + "_bench_kernels.json")))
rl_path = os.path.join(ROOT, "profiles", tag + "_roofline.json")
calls = json.load(open(rl_path))["calls"] if os.path.exists(rl_path) else {}


def peak_of(r):
    return "%g %s" % (r.get("peak", 0), r.get("unit", ""))


def fmt(x, nd=3):
    return "--" if x is None else ("%.*f" % (nd, x))


lines = ["| kernel / call (shape) | bound | us (bench) | achieved | frac of peak | rocprof us | counter traffic / algorithmic |",
         "|---|---|---|---|---|---|---|"]
order = sorted(rows.items(), key=lambda kv: -kv[1].get("us", 0))
skip = ("iou_sweep[", "nms_sweep[")
for name, r in order:
    if name.startswith(skip):
        continue
    alg = r.get("hbm_alg_bytes")
    traffic = r.get("traffic")
    ratio = None
    if traffic and alg:
        ratio = traffic / alg
    elif traffic and r.get("unit") == "GB/s" and r.get("achieved") and r.get("us"):
        ratio = traffic / (r["achieved"] * 1e9 * r["us"] * 1e-6)
    # the rocprof row of the same call, matched on the leading identifier
    key = name.split("(")[0].split("<")[0].strip()
    rp = next((v for k, v in calls.items() if key and key.split("_kernel")[0] in k.replace(" ", "_")), None)
    lines.append("| `%s` | %s | %.1f | %.4g %s | **%.3f** | %s | %s |" % (
        name.replace("|", "/"), r.get("bound", ""), r.get("us", 0), r.get("achieved", 0), r.get("unit", ""), r.get("frac", 0),
        fmt(rp["us"], 1) if rp else "--", fmt(ratio, 2) + " x" if ratio else "--"))
lines.append("")
lines.append("§8(d) sweeps (same file): " + "; ".join(
    "%s %.1f us (%.3f)" % (n, r["us"], r["frac"]) for n, r in rows.items() if n.startswith(skip)))
text = "\n".join(lines) + "\n"
open(os.path.join(ROOT, "profiles", tag + "_bounds_table.md"), "w").write(text)
dpath = os.path.join(ROOT, "DESIGN.md")
d = open(dpath).read()
a, b = "<!-- BOUNDS_TABLE -->", "<!-- /BOUNDS_TABLE -->"
if a in d:
    i = d.index(a) + len(a)
    j = d.index(b) if b in d else i
    d = d[:i] + "\n" + text + b + d[(j + len(b)) if b in d[j:j + len(b) + 1] or d[j:j + len(b)] == b else j:]
    open(dpath, "w").write(d)
print("%d rows -> profiles/%s_bounds_table.md" % (len(lines) - 3, tag))
