"""The two-tier dense rotated IoU at the S2ANet step shape, a few eager calls (for rocprofv3 --kernel-trace / --pmc)."""
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
ro = torch.tensor(np.concatenate([[0], np.cumsum(ks)]), dtype=torch.int32, device=dev)
grid = torch.from_numpy(syn.s2anet_anchor_grid()).to(dev)
ov = torch.empty((gt.shape[0], grid.shape[0]), device=dev)
prep = ops.prepare_boxes(grid, heavy_from=int(os.environ.get('HEAVY', 20480)))
pgt = ops.prepare_boxes(gt)
for _ in range(int(os.environ.get("REPS", "20"))):
    ops.box_iou_rotated_fast(gt, grid, ro, ks=ks, out=ov, prepared=prep, prepared1=pgt)
torch.cuda.synchronize()
