"""Per-workgroup stage timestamps of iou_fast_tile_kernel (debug build: profiles/scripts/ab_build.sh fasttrace
-DRSDET_FAST_TRACE):  RSDET_LIB_PATH=scratch/lib_fasttrace.so python profiles/scripts/trace_fast.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rs_detection_amd import _lib, ops  # noqa: E402
from rs_detection_amd.utils import synthetic as syn  # noqa: E402

dev = torch.device("cuda")
lib = _lib.load()
ks = [16, 100, 400, 40]
tg = syn.synthetic_targets(4)
gt = torch.cat([torch.from_numpy(t["rboxes"]) for t in tg]).to(dev)
ro = torch.tensor(np.concatenate([[0], np.cumsum(ks)]), dtype=torch.int32, device=dev)
grid = torch.from_numpy(syn.s2anet_anchor_grid()).to(dev)
prep = ops.prepare_boxes(grid, heavy_from=int(os.environ.get('HEAVY', 20480)))
pgt = ops.prepare_boxes(gt)
n1, A = gt.shape[0], grid.shape[0]
ov = torch.empty((n1, A), device=dev)
R = lib.rsdet_box_iou_rotated_fast_rows_per_tile()
nb = ((A + 255) // 256) * sum((k + R - 1) // R for k in ks) * 4
tr = torch.zeros(nb * 8, dtype=torch.int64, device=dev)
call = lambda: ops.box_iou_rotated_fast(gt, grid, ro, ks=ks, out=ov, prepared=prep, prepared1=pgt)
for _ in range(5):
    call()
torch.cuda.synchronize()
lib.rsdet_debug_set_fast_trace.argtypes = [ctypes.c_void_p]
lib.rsdet_debug_set_fast_trace(ctypes.c_void_p(tr.data_ptr()))
torch.cuda.synchronize()
call()
torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(nb, 8).astype(np.float64) * 0.01
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
last = t[:, :7].max(axis=1)
print("blocks", len(t), "kernel span %.2f us" % (last.max() - t0))
print("block start pct 10/50/90/100 =", np.percentile(t[:, 0] - t0, [10, 50, 90, 100]).round(2))
names = ["fill issue + stage + barrier", "cull + circles + scan", "separating axes + scan", "wait for the zero stores",
         "tier 1 (Green)", "tier 2 (reference clipper)"]
for k in range(6):
    ok = (t[:, k + 1] > 0) & (t[:, k] > 0)
    d = t[ok, k + 1] - t[ok, k]
    if ok.sum():
        print("%-30s n=%d mean %.2f p50 %.2f p90 %.2f max %.2f" % (names[k], ok.sum(), d.mean(), *np.percentile(d, [50, 90, 100])))
print("block end pct 10/50/90/100 =", np.percentile(last - t0, [10, 50, 90, 100]).round(2))
life = last - t[:, 0]
print("block lifetime mean %.2f p50 %.2f p90 %.2f max %.2f; sum %.0f us" % (life.mean(), *np.percentile(life, [50, 90, 100]), life.sum()))
order = np.argsort(-last)[:14]
print("slowest blocks: start, end | stage durations (a zero stamp = stage not reached)")
for o in order:
    print("%.1f %.1f |" % (t[o, 0] - t0, last[o] - t0), np.round(np.diff(np.where(t[o, :7] > 0, t[o, :7], np.nan)), 1))
print("histogram of block ends (us):", np.histogram(last - t0, bins=[0, 6, 8, 10, 12, 14, 16, 18, 20, 22, 30])[0])
