"""The rotated-IoU forms at the S2ANet step shape, a few eager calls each (for rocprofv3 --kernel-trace / --pmc)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rs_detection_amd import ops  # noqa: E402
from rs_detection_amd.utils import synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
ks = [16, 100, 400, 40]
tg = syn.synthetic_targets(4)
gt = torch.cat([torch.from_numpy(t["rboxes"]) for t in tg]).to(dev)
lab = torch.cat([torch.from_numpy(t["labels"]) for t in tg]).to(dev).int()
ro = torch.tensor(np.concatenate([[0], np.cumsum(ks)]), dtype=torch.int32, device=dev)
grid = torch.from_numpy(syn.s2anet_anchor_grid()).to(dev)
ov = torch.empty((gt.shape[0], grid.shape[0]), device=dev)
which = sys.argv[1:] or ["r1", "tiled", "fused"]
n = int(os.environ.get("REPS", "10"))
prep = ops.prepare_boxes(grid, heavy_from=int(os.environ.get('HEAVY', 20480)))
pgt = ops.prepare_boxes(gt)
for _ in range(n):
    if "r1" in which:
        ops.box_iou_rotated_grouped(gt, ro, max(ks), grid, out=ov)
        ops.assign_wrt_overlaps(ov, ro, max(ks), 0.5, 0.4, 0.0, True, True, lab, 0)
    if "tiled" in which:
        ops.box_iou_rotated_tiled(gt, grid, ro, ks=ks, out=ov, prepared=prep, prepared1=pgt, split=False)
    if "split" in which:
        ops.box_iou_rotated_tiled(gt, grid, ro, ks=ks, out=ov, prepared=prep)
    if "fused" in which:
        ops.anchor_target_rotated(grid, gt, lab, ro, ks, 0.5, 0.4, 0.0, prepared=prep, prepared_gt=pgt)
torch.cuda.synchronize()
