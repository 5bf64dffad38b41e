"""Who launches what: every GPU kernel of a train step attributed to the aten / autograd operator and the innermost
rs_detection_amd Python frame that launched it (torch.profiler trace, correlation ids).  Answers "where do the 66
bfloat16_copy kernels per step come from" -- the per-kernel tables of rocprofv3 cannot.

    python3 profiles/scripts/kernel_attrib.py [--dtype bf16] [--model s2anet_r50] [--steps 4] [--filter copy,add,reduce]

Timing under the profiler is NOT representative (host overhead of with_stack); only counts and attributions are used."""
import argparse
import bisect
import collections
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--model", default="s2anet_r50", choices=["s2anet_r50", "s2anet_r101", "orcnn_van3"])
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--filter", default="")
    ap.add_argument("--top", type=int, default=70)
    args = ap.parse_args()
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
    import torch
    from rs_detection_amd.utils.miopen_db import use_packaged_miopen_db
    use_packaged_miopen_db()
    import bench
    from rs_detection_amd.runner.runner import Runner
    from rs_detection_amd.utils import synthetic as syn
    dev = torch.device("cuda", 0)
    b16 = args.dtype == "bf16"
    orcnn = args.model == "orcnn_van3"
    if orcnn:
        from rs_detection_amd.config import Config
        cfg, batch, ncls = Config(os.path.join(ROOT, "configs", "orcnn", "orcnn_van3_7_anchor.py")), 2, 10
    else:
        cfg, batch, ncls = bench.s2anet_cfg(), 4, 15
        if args.model == "s2anet_r101":
            cfg = Config(os.path.join(ROOT, "configs", "s2anet", "s2anet_r101_fpn_1x_dota_rotate_balance_ms.py"))
    mf = None if orcnn else torch.channels_last
    runner = Runner(cfg, device=dev, amp_dtype=torch.bfloat16 if b16 else None, memory_format=mf, bf16_params=b16)
    batches = bench.make_batches(2, batch, 0, ncls, dev, mf, orcnn)
    inner = runner

    class _R:
        def train_step(self, b):
            return inner.train_step(*b)
    runner = _R()
    for i in range(6):
        runner.train_step(batches[i % len(batches)])
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for i in range(args.steps):
            runner.train_step(batches[i % len(batches)])
        torch.cuda.synchronize()
    path = os.path.join(tempfile.gettempdir(), "attrib_trace.json")
    prof.export_chrome_trace(path)
    ev = json.load(open(path))["traceEvents"]
    kernels = [e for e in ev if e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset")]
    launches = {e["args"]["correlation"]: e for e in ev
                if e.get("cat") in ("cuda_runtime", "cuda_driver") and "correlation" in e.get("args", {})}
    # per thread: sorted intervals of cpu_op and python_function events
    per_tid = collections.defaultdict(lambda: {"cpu_op": [], "python_function": []})
    for e in ev:
        if e.get("cat") in ("cpu_op", "python_function") and "dur" in e:
            per_tid[e["tid"]][e["cat"]].append((e["ts"], e["ts"] + e["dur"], e["name"]))
    for d in per_tid.values():
        for k in d:
            d[k].sort()

    def enclosing(lst, ts, pred=None, outer=False):
        """names of intervals containing ts (outermost first)"""
        i = bisect.bisect_right(lst, (ts, float("inf"), ""))
        out = []
        j = i - 1
        while j >= 0 and len(out) < 64:
            a, b, n = lst[j]
            if a <= ts <= b and (pred is None or pred(n)):
                out.append(n)
            if ts - a > 5e6:
                break
            j -= 1
        return out[::-1]

    table = collections.defaultdict(lambda: [0, 0.0])
    for k in kernels:
        corr = k.get("args", {}).get("correlation")
        la = launches.get(corr)
        op, frame = "?", "?"
        if la is not None:
            d = per_tid[la["tid"]]
            ops = enclosing(d["cpu_op"], la["ts"])
            if ops:
                top = ops[0].replace("autograd::engine::evaluate_function: ", "bwd:")
                op = top if len(ops) == 1 else top + " > " + ops[-1]
            fr = enclosing(d["python_function"], la["ts"], lambda n: "rs_detection_amd" in n or "bench.py" in n)
            if fr:
                frame = fr[-1].split("rs_detection_amd/")[-1]
        name = k["name"]
        short = name.split("(")[0][-70:] if len(name) > 70 else name
        key = (short, op[:80], frame[:70])
        table[key][0] += 1
        table[key][1] += k.get("dur", 0.0)
    flt = [f for f in args.filter.split(",") if f]
    rows = sorted(table.items(), key=lambda kv: -kv[1][1])
    print("calls/step   us/step  kernel | operator | python frame")
    n = 0
    for (short, op, frame), (cnt, dur) in rows:
        if flt and not any(f in short for f in flt):
            continue
        print("%8.1f %9.1f  %s | %s | %s" % (cnt / args.steps, dur / args.steps, short, op, frame))
        n += 1
        if n >= args.top:
            break
    # the same launches by Python frame (library + torch kernels only): where a fused kernel would remove the most launches
    by = collections.defaultdict(lambda: [0, 0.0])
    for (short, op, frame), (cnt, dur) in rows:
        if "rsdet::" in short:
            continue
        key = frame if frame != "?" else op.split(" > ")[0]
        by[key][0] += cnt
        by[key][1] += dur
    print("\ncalls/step   us/step  python frame (or autograd node) -- kernels that are not rsdet::")
    for key, (cnt, dur) in sorted(by.items(), key=lambda kv: -kv[1][0])[:60]:
        print("%8.1f %9.1f  %s" % (cnt / args.steps, dur / args.steps, key))


if __name__ == "__main__":
    main()
