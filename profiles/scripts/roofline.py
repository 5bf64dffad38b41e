"""Merge one rocprofv3 kernel trace, the two PMC passes and the code-object notes into the roofline table:
    python profiles/scripts/roofline.py <dir with trace/ pmc_WRITE_SIZE/ pmc_FETCH_SIZE/ alg_bytes.json> out.json
Per hand-written kernel: launches, average / min duration (us), HBM bytes per launch from the counters (FETCH_SIZE is
reported in KB and counts 64 B per 128-B request on gfx950 for wide streaming reads: the "x2" figure is given beside the
raw one, MI355X_MICROARCH.md), VGPR / SGPR / LDS / scratch and waves per SIMD from the code objects
(profiles/scripts/occupancy.py).  Per C-ABI call (the launches that share one algorithmic-byte figure):
sum of the average durations, algorithmic bytes, achieved GB/s and the fraction of the 8 TB/s HBM peak."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import occupancy  # noqa: E402

HBM_PEAK = 8000.0  # GB/s


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main(d, out):
    trace = glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True)[0]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        n = short(r["Kernel_Name"])
        if "rsdet::" in n:
            dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    pmc = {}
    for c in ("WRITE_SIZE", "FETCH_SIZE"):
        acc = collections.defaultdict(list)
        for f in glob.glob(os.path.join(d, "pmc_" + c, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)   # KB -> bytes
        pmc[c] = {k: sum(v) / len(v) for k, v in acc.items()}
    occ = occupancy.table()
    alg = json.load(open(os.path.join(d, "alg_bytes.json")))
    kernels, calls = {}, collections.OrderedDict()
    for n, v in sorted(dur.items()):
        v = sorted(v)
        # drop the first launch of each kernel (code-object load / cold caches) when there are several
        vv = v if len(v) < 3 else v[:-1] if v[-1] > 3 * v[len(v) // 2] else v
        row = dict(launches=len(v), avg_us=sum(vv) / len(vv), min_us=v[0],
                   write_bytes=pmc["WRITE_SIZE"].get(n), fetch_bytes_reported=pmc["FETCH_SIZE"].get(n))
        if row["fetch_bytes_reported"] is not None:
            row["fetch_bytes_x2"] = 2 * row["fetch_bytes_reported"]
            row["traffic_bytes"] = (row["write_bytes"] or 0) + row["fetch_bytes_x2"]
        o = occ.get(n.replace("rsdet::", "rsdet::")) or next((r for k, r in occ.items() if k == n), None)
        if o:
            row.update({k: o[k] for k in ("vgpr", "agpr", "sgpr", "lds_bytes", "scratch_bytes", "waves_per_simd")})
        key = next((k for k in alg if k in n), None)
        if key:
            row["call"] = alg[key]["call"]
            c = calls.setdefault(alg[key]["call"], dict(alg_bytes=alg[key]["bytes"], kernels=[], us=0.0, traffic_bytes=0.0))
            if "flops" in alg[key]:
                c["alg_flops"] = alg[key]["flops"]
            c["kernels"].append(n)
            c["us"] += row["avg_us"]
            c["traffic_bytes"] += row.get("traffic_bytes") or 0.0
        kernels[n] = row
    for c in calls.values():
        c["achieved_GBps"] = c["alg_bytes"] / (c["us"] * 1e-6) / 1e9
        c["frac_of_hbm_peak"] = c["achieved_GBps"] / HBM_PEAK
        c["traffic_over_alg"] = c["traffic_bytes"] / c["alg_bytes"] if c["alg_bytes"] else None
        if "alg_flops" in c:
            c["achieved_TFLOPs"] = c["alg_flops"] / (c["us"] * 1e-6) / 1e12
    issue = issue_side(d, calls, kernels)
    res = dict(hbm_peak_GBps=HBM_PEAK, source="rocprofv3 --kernel-trace + --pmc WRITE_SIZE / FETCH_SIZE (+ two SQ counter "
               "passes) over profiles/scripts/pmc_kernels.py; code-object notes via profiles/scripts/occupancy.py",
               calls=calls, kernels=kernels, issue=issue)
    json.dump(res, open(out, "w"), indent=1)
    for name, c in calls.items():
        print("%-44s %8.1f us  alg %8.2f MB  %7.0f GB/s  frac %.3f  traffic/alg %.2f" % (
            name, c["us"], c["alg_bytes"] / 1e6, c["achieved_GBps"], c["frac_of_hbm_peak"], c["traffic_over_alg"] or 0))


N_SIMD, CLK_GHZ = 1024, 2.4      # 256 CUs x 4 SIMDs; nominal shader clock (MI355X_MICROARCH.md)
ISSUE_CALLS = {"anchor_target": "anchor_target_rotated (2 launches)", "nms_rotated": "nms_rotated (3 launches)",
               "dense_iou_two_tier": "box_iou_rotated_fast (1 launch)", "dense_iou_exact": "box_iou_rotated (3 launches)"}


def issue_side(d, calls, kernels):
    """The VALU / latency side of the calls whose HBM fraction says nothing: per call, summed over its kernels,
    VALU instructions issued (SQ_INSTS_VALU), the share of the launch's VALU issue slots they fill -- one slot = one
    quad-cycle of one SIMD (a wave's VALU instruction holds its SIMD's issue port for 4 clocks: SQ_ACTIVE_INST_VALU ~
    SQ_INSTS_VALU quad-cycles), slots = 1024 SIMDs x duration x 2.4 GHz / 4 -- and the share of wave-cycles spent
    waiting (s_waitcnt / barrier) or stalled at issue."""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "sq*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc:
        return {}
    out = {}
    for tag, call in ISSUE_CALLS.items():
        c = calls.get(call)
        if not c:
            continue
        tot = collections.defaultdict(float)
        for k in c["kernels"]:
            for name, v in acc.get(k, {}).items():
                tot[name] += sum(v) / len(v)
        if not tot.get("SQ_INSTS_VALU"):
            continue
        slots = N_SIMD * c["us"] * 1e-6 * CLK_GHZ * 1e9 / 4.0
        wc = tot.get("SQ_WAVE_CYCLES") or 1.0
        out[tag] = dict(call=call, us=c["us"], waves=tot.get("SQ_WAVES"), valu_insts=tot["SQ_INSTS_VALU"],
                        valu_issue_frac=tot["SQ_INSTS_VALU"] / slots, salu_insts=tot.get("SQ_INSTS_SALU"),
                        lds_insts=tot.get("SQ_INSTS_LDS"), wait_frac=tot.get("SQ_WAIT_ANY", 0.0) / wc,
                        issue_stall_frac=tot.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                        note="valu_issue_frac = VALU instructions / (1024 SIMDs x duration x 2.4 GHz / 4)")
    return out


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
