"""Per-workgroup stage timestamps of iou_tile_kernel (debug build -DRSDET_TILE_TRACE, scratch/lib_tiletrace.so):
RSDET_LIB_PATH=scratch/lib_tiletrace.so python profiles/scripts/trace_tiles.py [dense|sparse]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rs_detection_amd import _lib, ops  # noqa: E402
from rs_detection_amd.utils import synthetic as syn  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "dense"
dev = torch.device("cuda")
lib = _lib.load()
ks = [16, 100, 400, 40]
tg = syn.synthetic_targets(4)
gt = torch.cat([torch.from_numpy(t["rboxes"]) for t in tg]).to(dev)
lab = torch.cat([torch.from_numpy(t["labels"]) for t in tg]).to(dev).int()
ro = torch.tensor(np.concatenate([[0], np.cumsum(ks)]), dtype=torch.int32, device=dev)
grid = torch.from_numpy(syn.s2anet_anchor_grid()).to(dev)
prep = ops.prepare_boxes(grid, heavy_from=int(os.environ.get('HEAVY', 20480)))
pgt = ops.prepare_boxes(gt) if os.environ.get("PREP_GT", "1") == "1" else None
n1, A = gt.shape[0], grid.shape[0]
ov = torch.empty((n1, A), device=dev)
nrt = sum((k + 15) // 16 for k in ks)
nb = ((A + 255) // 256) * nrt * 4          # upper bound of the grid (every column tile cut into 4 sub-tiles)
tr = torch.zeros(nb * 8, dtype=torch.int64, device=dev)


def call():
    if mode == "dense":
        ops.box_iou_rotated_tiled(gt, grid, ro, ks=ks, out=ov, prepared=prep, prepared1=pgt)
    else:
        ops.anchor_target_rotated(grid, gt, lab, ro, ks, 0.5, 0.4, 0.0, prepared=prep, prepared_gt=pgt)


for _ in range(5):
    call()
torch.cuda.synchronize()
lib.rsdet_debug_set_tile_trace.argtypes = [ctypes.c_void_p]
lib.rsdet_debug_set_tile_trace(ctypes.c_void_p(tr.data_ptr()))
torch.cuda.synchronize()
call()
torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(nb, 8).astype(np.float64) * 0.01
t = t[t[:, 0] > 0]
nb = len(t)
t0 = t[:, 0].min()
last = t[:, :6].max(axis=1)
print(mode, "blocks", nb, "kernel span %.2f us" % (last.max() - t0))
print("block start pct 10/50/90/100 =", np.percentile(t[:, 0] - t0, [10, 50, 90, 100]).round(2))
names = ["stage (copy + rows + barrier)", "cull + circles", "separating axes", "fill issue / reserve", "clip"]
for k in range(5):
    ok = (t[:, k + 1] > 0) & (t[:, k] > 0)
    d = t[ok, k + 1] - t[ok, k]
    if ok.sum():
        print("%-30s n=%d mean %.2f p50 %.2f p90 %.2f max %.2f" % (names[k], ok.sum(), d.mean(), *np.percentile(d, [50, 90, 100])))
print("block end pct 10/50/90/100 =", np.percentile(last - t0, [10, 50, 90, 100]).round(2))
life = last - t[:, 0]
print("block lifetime mean %.2f p50 %.2f p90 %.2f max %.2f; sum %.0f us" % (life.mean(), *np.percentile(life, [50, 90, 100]), life.sum()))
