"""Micro-benchmark of the rotated-IoU / anchor-target forms at the S2ANet step shape (HIP events around a replayed
hipGraph, bench.event_time).  Usage: python profiles/scripts/iou_bench.py [--refined]"""
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
    lab = torch.cat([torch.from_numpy(t["labels"]) for t in tg]).to(dev).int()
    ro = torch.tensor(np.concatenate([[0], np.cumsum(ks)]), dtype=torch.int32, device=dev)
    grid = torch.from_numpy(syn.s2anet_anchor_grid()).to(dev)
    rng = np.random.default_rng(7)
    ref = np.stack([syn.refined_anchor_grid(seed=7 + i) for i in range(4)])
    refined = torch.from_numpy(ref).to(dev)
    n1, A = gt.shape[0], grid.shape[0]
    out = {}
    for name, anchors in (("grid", grid), ("refined(B,A,5)", refined)):
        ov = torch.empty((n1, A), device=dev)
        prep = ops.prepare_boxes(anchors, heavy_from=int(os.environ.get('HEAVY', 20480)))
        r = {}
        r["r1 three launches (prepare+filter+clip)"] = event_time(
            lambda: ops.box_iou_rotated_grouped(gt, ro, max(ks), anchors, out=ov), 50) * 1e6
        pgt = ops.prepare_boxes(gt)
        r["split (detect | fill+clip), prepared anchors cached (2 launches)"] = event_time(
            lambda: ops.box_iou_rotated_tiled(gt, anchors, ro, ks=ks, out=ov, prepared=prep), 50) * 1e6
        r["split, prepared anchors + gts cached"] = event_time(
            lambda: ops.box_iou_rotated_tiled(gt, anchors, ro, ks=ks, out=ov, prepared=prep, prepared1=pgt), 50) * 1e6
        r["split, prepare every call (3 launches)"] = event_time(
            lambda: ops.box_iou_rotated_tiled(gt, anchors, ro, ks=ks, out=ov), 50) * 1e6
        r["split, no tile table"] = event_time(
            lambda: ops.box_iou_rotated_tiled(gt, anchors, ro, max_rows=max(ks), out=ov, prepared=prep), 50) * 1e6
        r["one launch (tile finished by its workgroup), prepared cached"] = event_time(
            lambda: ops.box_iou_rotated_tiled(gt, anchors, ro, ks=ks, out=ov, prepared=prep, split=False), 50) * 1e6
        r["prepare only"] = event_time(lambda: ops.prepare_boxes(anchors), 50) * 1e6
        r["r1 assign (row+col)"] = event_time(
            lambda: ops.assign_wrt_overlaps(ov, ro, max(ks), 0.5, 0.4, 0.0, True, True, lab, 0), 50) * 1e6
        r["fused anchor_target (cached prepare; 2 launches)"] = event_time(
            lambda: ops.anchor_target_rotated(anchors, gt, lab, ro, ks, 0.5, 0.4, 0.0, prepared=prep), 50) * 1e6
        r["fused anchor_target (prepare every call; 3 launches)"] = event_time(
            lambda: ops.anchor_target_rotated(anchors, gt, lab, ro, ks, 0.5, 0.4, 0.0), 50) * 1e6
        by = 20 * (n1 + A) + 4 * n1 * A
        r["alg MB"] = by / 1e6
        out[name] = {k: round(v, 2) for k, v in r.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
