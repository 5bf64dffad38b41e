"""Kernel-time breakdown of a bench.py run under rocprofv3 --kernel-trace --stats: per family (MIOpen conv, rocBLAS /
Tensile, MIOpen transposes, torch elementwise / reduce / copy, hand-written rsdet) ms per step + the top kernels.
usage: step_breakdown.py <dir> <steps counted in the run (warmup + flop step + timed)>"""
import csv, glob, sys, collections
d, steps = sys.argv[1], float(sys.argv[2])
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
# steps = 0: count them from a kernel that runs a known number of times per step (the VAN-B3 backbone: 38 blocks, one
# rsdet::van_transposes_kernel launch each) -- the bench's set-up steps make the nominal count wrong
if steps == 0:
    for r in csv.DictReader(open(f)):
        if "van_transposes_kernel" in r["Name"]:
            steps = int(r["Calls"]) / 38.0
fam = collections.defaultdict(float)
rows = []
for r in csv.DictReader(open(f)):
    n, t, c = r["Name"], float(r["TotalDurationNs"]) / 1e6, int(r["Calls"])
    rows.append((t, c, n))
    if "rsdet::" in n: k = "hand-written (rsdet)"
    elif "batched_transpose" in n or "SubTensorOp" in n: k = "MIOpen layout transposes / subtensor"
    elif n.startswith("Cijk_") or "rocblas" in n.lower(): k = "rocBLAS / Tensile GEMM"
    elif "miopen" in n.lower() or "igemm" in n or "Conv" in n or "conv" in n or "ck::" in n or "kernel_grouped_conv" in n: k = "MIOpen / CK convolution"
    elif "at::native" in n or "at_cuda" in n: k = "torch elementwise / reduce / copy"
    else: k = "other"
    fam[k] += t
tot = sum(fam.values())
print("kernel time per step: %.2f ms over %.0f steps" % (tot / steps, steps))
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
    print("  %-42s %7.2f ms/step" % (k, v / steps))
for t, c, n in sorted(rows, reverse=True)[:int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    print("  %7.3f ms/step %6.0f calls/step  %s" % (t / steps, c / steps, n[:100]))
