"""Device time of the dense rotated-IoU forms at the S2ANet step shape (HIP events around a replayed hipGraph).
Usage: python profiles/scripts/iou_fast_bench.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import event_time  # noqa: E402
from rs_detection_amd import ops  # noqa: E402
from rs_detection_amd.utils import synthetic as syn  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ks = [16, 100, 400, 40]
    tg = syn.synthetic_targets(4)
    gt = torch.cat([torch.from_numpy(t["rboxes"]) for t in tg]).to(dev)
    ro = torch.tensor(np.concatenate([[0], np.cumsum(ks)]), dtype=torch.int32, device=dev)
    grid = torch.from_numpy(syn.s2anet_anchor_grid()).to(dev)
    refined = torch.from_numpy(np.stack([syn.refined_anchor_grid(seed=7 + i) for i in range(4)])).to(dev)
    n1, A = gt.shape[0], grid.shape[0]
    by = 20 * (n1 + A) + 4 * n1 * A
    out = {}
    for name, anchors in (("grid", grid), ("refined(B,A,5)", refined)):
        ov = torch.empty((n1, A), device=dev)
        prep = ops.prepare_boxes(anchors, heavy_from=int(os.environ.get('HEAVY', 20480)))
        pgt = ops.prepare_boxes(gt)
        r = {}
        r["exact, three launches (prepare+filter+clip)"] = event_time(
            lambda: ops.box_iou_rotated_grouped(gt, ro, max(ks), anchors, out=ov), 50) * 1e6
        exact = ov.clone()
        r["two-tier, prepared anchors + gts given (1 launch)"] = event_time(
            lambda: ops.box_iou_rotated_fast(gt, anchors, ro, ks=ks, out=ov, prepared=prep, prepared1=pgt), 50) * 1e6
        r["two-tier, prepared anchors given, gts prepared in the tile (1 launch)"] = event_time(
            lambda: ops.box_iou_rotated_fast(gt, anchors, ro, ks=ks, out=ov, prepared=prep), 50) * 1e6
        r["two-tier, prepare every call (2 launches)"] = event_time(
            lambda: ops.box_iou_rotated_fast(gt, anchors, ro, ks=ks, out=ov), 50) * 1e6
        r["two-tier, no tile table, prepare every call"] = event_time(
            lambda: ops.box_iou_rotated_fast(gt, anchors, ro, max_rows=max(ks), out=ov), 50) * 1e6
        lab = torch.cat([torch.from_numpy(t["labels"]) for t in tg]).to(dev).int()
        for tt in (False, True):
            tag = "two-tier" if tt else "exact"
            r["fused anchor targets, %s (cached prepare; 2 launches)" % tag] = event_time(
                lambda: ops.anchor_target_rotated(anchors, gt, lab, ro, ks, 0.5, 0.4, 0.0, prepared=prep, prepared_gt=pgt,
                                                  two_tier=tt), 50) * 1e6
            r["fused anchor targets, %s (anchors cached, gts prepared in the tiles)" % tag] = event_time(
                lambda: ops.anchor_target_rotated(anchors, gt, lab, ro, ks, 0.5, 0.4, 0.0, prepared=prep, two_tier=tt), 50) * 1e6
        r["max |two-tier - exact|"] = float((ov - exact).abs().max())
        r["zeros agree"] = bool(((ov == 0) == (exact == 0)).all())
        r["memset of the matrix (torch.zero_)"] = event_time(lambda: ov.zero_(), 50) * 1e6
        r["alg MB"] = by / 1e6
        for k in list(r):
            if isinstance(r[k], float) and "launch" in k:
                r[k + " [frac of 8 TB/s]"] = round(by / (r[k] * 1e-6) / 8e12, 3)
        out[name] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
