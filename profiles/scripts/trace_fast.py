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
nb = 16384          # more than the launch has workgroups (store + compute)
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
raw = tr.cpu().numpy().reshape(nb, 8)
info = raw[:, 4].copy()
raw[:, 4] = 0
t = raw.astype(np.float64) * 0.01
used = t[:, 0] > 0
info = info[used]
bid = np.nonzero(used)[0]
t = t[used]
t0 = t[:, 0].min()
last = t[:, :8].max(axis=1)
print("blocks", len(t), "kernel span %.2f us" % (last.max() - t0))
# a store workgroup stamps 0 (start), 1 (cull done), 2 (zeros issued) only; a compute workgroup stamps 3 as well
store = (t[:, 3] == 0) & (t[:, 2] > 0) & (t[:, 7] == 0)
empty = (t[:, 1] == 0)
comp = ~store & ~empty
print("store workgroups %d, compute %d, empty (exit at once) %d" % (store.sum(), comp.sum(), empty.sum()))
for name, sel in (("store", store), ("compute", comp)):
    if not sel.sum():
        continue
    ts, ls = t[sel], last[sel]
    print("== %s: start pct 10/50/90/100 = %s; end pct 10/50/90/100 = %s" % (
        name, np.percentile(ts[:, 0] - t0, [10, 50, 90, 100]).round(2), np.percentile(ls - t0, [10, 50, 90, 100]).round(2)))
    stages = ([("loads + cull", 0, 1), ("zero stores issued", 1, 2)] if name == "store" else
              [("loads + stage + barrier", 0, 1), ("cull + circles + separating axes", 1, 2), ("scan", 2, 3),
               ("tier 1 (Green) + its stores", 3, 5), ("tier 2 (reference clipper)", 5, 6),
               ("zeros of the live cells issued", 6, 7)])
    for nm, k0, k1 in stages:
        ok = (ts[:, k1] > 0) & (ts[:, k0] > 0)
        d = ts[ok, k1] - ts[ok, k0]
        if ok.sum():
            print("  %-34s n=%d mean %.2f p50 %.2f p90 %.2f max %.2f" % (nm, ok.sum(), d.mean(), *np.percentile(d, [50, 90, 100])))
    life = ls - ts[:, 0]
    print("  lifetime mean %.2f p50 %.2f p90 %.2f max %.2f; sum %.0f us" % (life.mean(), *np.percentile(life, [50, 90, 100]), life.sum()))
    order = np.argsort(-ls)[:12]
    print("  slowest: start, end | stamps 0..7 (us since the first block; nan = not reached)")
    for o in order:
        io = int(info[sel][o])
        print("   block %d coltile %d survivors %d live0 %d | %.1f %.1f |" % (bid[sel][o], io >> 32, io & 0xffff, (io >> 16) & 0xffff,
              ts[o, 0] - t0, ls[o] - t0), np.round(np.where(ts[o, :8] > 0, ts[o, :8] - t0, np.nan), 1))
print("histogram of block ends (us):", np.histogram(last - t0, bins=[0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 30])[0])
ct = (info[comp] >> 32).astype(int)
lc = last[comp] - t0
dt = t[comp][:, 2] - t[comp][:, 1]
for lo, hi, nm in ((0, 64, "level 0"), (64, 80, "level 1"), (80, 84, "level 2"), (84, 85, "level 3"), (85, 86, "level 4")):
    m = (ct >= lo) & (ct < hi)
    if m.sum():
        print("%s: %d workgroups, survivors mean %.0f max %d, detect mean %.2f max %.2f, end mean %.2f max %.2f" % (
            nm, m.sum(), (info[comp][m] & 0xffff).mean(), (info[comp][m] & 0xffff).max(), dt[m].mean(), dt[m].max(), lc[m].mean(), lc[m].max()))
