"""Device time of the two-tier dense rotated IoU at the S2ANet step shape (HIP events around a replayed hipGraph) and its
agreement with the bit-exact op.  RSDET_LIB_PATH selects the build.  Usage: python profiles/scripts/iou_fast_quick.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import event_time  # noqa: E402
from rs_detection_amd import ops  # noqa: E402
from rs_detection_amd.utils import synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
ks = [16, 100, 400, 40]
tg = syn.synthetic_targets(4)
gt = torch.cat([torch.from_numpy(t["rboxes"]) for t in tg]).to(dev)
ro = torch.tensor(np.concatenate([[0], np.cumsum(ks)]), dtype=torch.int32, device=dev)
grid = torch.from_numpy(syn.s2anet_anchor_grid()).to(dev)
refined = torch.from_numpy(np.stack([syn.refined_anchor_grid(seed=7 + i) for i in range(4)])).to(dev)
n1, A = gt.shape[0], grid.shape[0]
by = 20 * (n1 + A) + 4 * n1 * A
for name, anchors in (("grid", grid), ("refined", refined)):
    ov = torch.full((n1, A), -7.0, device=dev)
    prep = ops.prepare_boxes(anchors, heavy_from=int(os.environ.get('HEAVY', 20480)))
    pgt = ops.prepare_boxes(gt)
    exact = ops.box_iou_rotated_grouped(gt, ro, max(ks), anchors)
    ops.box_iou_rotated_fast(gt, anchors, ro, ks=ks, out=ov, prepared=prep, prepared1=pgt)
    d = float((ov - exact).abs().max())
    z = bool(((ov == 0) == (exact == 0)).all())
    ts = sorted(event_time(lambda: ops.box_iou_rotated_fast(gt, anchors, ro, ks=ks, out=ov, prepared=prep, prepared1=pgt), 50) * 1e6
                for _ in range(5))
    print("%-8s %s  two-tier %.2f us (min of 5: %.2f) = %.3f of 8 TB/s | max diff %.2e zeros agree %s" % (
        name, os.environ.get("RSDET_LIB_PATH", "default"), ts[2], ts[0], by / (ts[2] * 1e-6) / 8e12, d, z))
