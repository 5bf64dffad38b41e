"""One process of tests/test_gpu_guards.py (not a test module): runs train steps of a BASELINE config at the bench's
shapes with the launch-counter shim preloaded (tests/tools/launch_counter.cpp) and reports, as one JSON line,
  * kernel launches / async fills / async copies PER STEP (whoever issued them: torch, MIOpen, BLAS, librsdet_hip.so),
  * the host<->device synchronisations of one step (torch.cuda.set_sync_debug_mode: "error" for the S2ANet steps, which
    must have none; "warn" + a count for Oriented R-CNN, whose samplers are data-dependent).
argv: model (s2anet | orcnn) dtype (f32 | bf16) tile"""
import ctypes
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    model, dtype, tile = sys.argv[1], sys.argv[2], int(sys.argv[3])
    lc = ctypes.CDLL(os.environ["RSDET_LAUNCH_COUNTER"])           # the preloaded shim: same handle, its counters
    for f in ("rsdet_lc_launches", "rsdet_lc_fills", "rsdet_lc_copies"):
        getattr(lc, f).restype = ctypes.c_longlong
    from rs_detection_amd.config import Config
    from rs_detection_amd.runner.runner import Runner
    from rs_detection_amd.utils import synthetic as syn
    dev = torch.device("cuda:0")
    orcnn = model == "orcnn"
    cfg = Config(os.path.join(ROOT, "configs", "orcnn", "orcnn_van3_7_anchor.py") if orcnn else
                 os.path.join(ROOT, "configs", "s2anet", "s2anet_r50_fpn_1x_dota.py"))
    batch, ncls = (2, 10) if orcnn else (4, 15)
    mf = None if orcnn else torch.channels_last
    torch.manual_seed(0)
    runner = Runner(cfg, device=dev, memory_format=mf, amp_dtype=torch.bfloat16 if dtype == "bf16" else None)
    batches = []
    for it in range(2):
        g = torch.Generator().manual_seed(it)
        im = torch.randn(batch, 3, tile, tile, generator=g).to(dev)
        if mf is not None:
            im = im.contiguous(memory_format=mf)
        tg = []
        for t in syn.synthetic_targets(batch, rank=0, it=it, num_classes=ncls, img=tile, k_shift=it):
            t = dict(t)
            t["rboxes"], t["labels"] = torch.from_numpy(t["rboxes"]).to(dev), torch.from_numpy(t["labels"]).to(dev)
            if orcnn:
                t["hboxes"] = None
            tg.append(t)
        batches.append((im, tg))
    for i in range(4):
        runner.train_step(*batches[i % 2])
    torch.cuda.synchronize()
    n = 4
    c0 = (lc.rsdet_lc_launches(), lc.rsdet_lc_fills(), lc.rsdet_lc_copies())
    for i in range(n):
        runner.train_step(*batches[i % 2])
    torch.cuda.synchronize()
    c1 = (lc.rsdet_lc_launches(), lc.rsdet_lc_fills(), lc.rsdet_lc_copies())
    per = [(b - a) / n for a, b in zip(c0, c1)]
    # -- synchronisations of one step
    syncs, err = 0, None
    if orcnn:
        torch.cuda.set_sync_debug_mode("warn")
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            runner.train_step(*batches[0])
        torch.cuda.set_sync_debug_mode("default")
        syncs = sum("synchroniz" in str(w.message).lower() for w in rec)
    else:
        torch.cuda.set_sync_debug_mode("error")
        try:
            loss, _ = runner.train_step(*batches[0])
        except RuntimeError as e:
            err = str(e)[:300]
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    print(json.dumps(dict(model=model, dtype=dtype, tile=tile, launches=per[0], fills=per[1], copies=per[2],
                          syncs=syncs, sync_error=err)))


if __name__ == "__main__":
    main()
