"""Device time of the grid-aware dense rotated IoU (csrc/iou_grid.hip) against the tile form (csrc/iou_fast.hip) at the
S2ANet step shape, for generated and refined anchors, with and without prepared gts; and that the two agree bit for
bit.  RSDET_LIB_PATH selects the build.  Usage: python profiles/scripts/iou_grid_quick.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import event_time  # noqa: E402
from rs_detection_amd import ops  # noqa: E402
from rs_detection_amd.ops.anchor_target import s2anet_grid_spec  # noqa: E402
from rs_detection_amd.utils import synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
ks = [16, 100, 400, 40]
tg = syn.synthetic_targets(4)
gt = torch.cat([torch.from_numpy(t["rboxes"]) for t in tg]).to(dev)
ro = torch.tensor(np.concatenate([[0], np.cumsum(ks)]), dtype=torch.int32, device=dev)
grid = torch.from_numpy(syn.s2anet_anchor_grid()).to(dev)
n1, A = gt.shape[0], grid.shape[0]
by = 20 * (n1 + A) + 4 * n1 * A
strides = (8, 16, 32, 64, 128)
sizes = [(1024 // s, 1024 // s) for s in strides]
tag = os.environ.get("RSDET_LIB_PATH", "default")
spec = s2anet_grid_spec(sizes, strides)
ov, ov2 = torch.full((n1, A), -7.0, device=dev), torch.full((n1, A), -7.0, device=dev)
prep = ops.prepare_boxes(grid, heavy_from=20480)
pgt = ops.prepare_boxes(gt)
for p1 in (None, pgt):
    ops.box_iou_rotated_fast(gt, grid, ro, ks=ks, out=ov, prepared=prep, prepared1=p1)
    ops.box_iou_rotated_grid(gt, grid, spec, out=ov2, prepared1=p1)
    same = bool((ov == ov2).all())
    tf = sorted(event_time(lambda: ops.box_iou_rotated_fast(gt, grid, ro, ks=ks, out=ov, prepared=prep, prepared1=p1), 50) * 1e6
                for _ in range(5))
    tgd = sorted(event_time(lambda: ops.box_iou_rotated_grid(gt, grid, spec, out=ov2, prepared1=p1), 50) * 1e6
                 for _ in range(5))
    print("grid %s gts %-8s tile form %.2f us (%.3f) | grid form %.2f us (min %.2f) = %.3f of 8 TB/s | identical %s" % (
        tag, "prepared" if p1 is not None else "raw", tf[2], by / (tf[2] * 1e-6) / 8e12, tgd[2], tgd[0],
        by / (tgd[2] * 1e-6) / 8e12, same))
# the SURVEY 8(d) micro-bench shapes: one group of K gts against the grid
for K in (16, 100, 400):
    g1 = torch.from_numpy(syn.dota_gt_boxes(np.random.default_rng(K), K)).to(dev)
    o = torch.empty((K, A), device=dev)
    b = 20 * (K + A) + 4 * K * A
    tf = event_time(lambda: ops.box_iou_rotated_fast(g1, grid, out=o, prepared=prep), 50) * 1e6
    tgd = event_time(lambda: ops.box_iou_rotated_grid(g1, grid, spec, out=o), 50) * 1e6
    print("K=%3d x A=%d: tile form %.2f us (%.3f) | grid form %.2f us (%.3f of 8 TB/s)" % (K, A, tf, b / tf / 8e6, tgd, b / tgd / 8e6))
