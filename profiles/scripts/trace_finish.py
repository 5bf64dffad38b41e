"""Stage timestamps of at_finish_kernel (debug build -DRSDET_TILE_TRACE -DRSDET_TRACE_FINISH_ONLY)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rs_detection_amd import _lib, ops
from rs_detection_amd.utils import synthetic as syn
dev = torch.device("cuda"); lib = _lib.load()
ks = [16, 100, 400, 40]
tg = syn.synthetic_targets(4)
gt = torch.cat([torch.from_numpy(t["rboxes"]) for t in tg]).to(dev)
lab = torch.cat([torch.from_numpy(t["labels"]) for t in tg]).to(dev).int()
ro = torch.tensor(np.concatenate([[0], np.cumsum(ks)]), dtype=torch.int32, device=dev)
grid = torch.from_numpy(syn.s2anet_anchor_grid()).to(dev)
prep = ops.prepare_boxes(grid, heavy_from=20480); pgt = ops.prepare_boxes(gt)
nb = 86 * 4
tr = torch.zeros(20000 * 8, dtype=torch.int64, device=dev)
call = lambda: ops.anchor_target_rotated(grid, gt, lab, ro, ks, 0.5, 0.4, 0.0, prepared=prep, prepared_gt=pgt)
for _ in range(5): call()
torch.cuda.synchronize()
lib.rsdet_debug_set_tile_trace.argtypes = [ctypes.c_void_p]
lib.rsdet_debug_set_tile_trace(ctypes.c_void_p(tr.data_ptr()))
torch.cuda.synchronize(); call(); torch.cuda.synchronize()
t = tr.cpu().numpy().reshape(-1, 8)[:nb].astype(np.float64) * 0.01
t0 = t[:, 0].min()
print("finish blocks", nb, "span %.2f us" % (t[:, :5].max() - t0), "starts p50/p100", np.percentile(t[:, 0] - t0, [50, 100]).round(2))
for k, n in enumerate(["first loads + barrier", "rowmax / dir scan / entries", "finalise (compute + stores)", "counts + done atomic"]):
    d = t[:, k + 1] - t[:, k]
    print("%-30s mean %.2f p50 %.2f p90 %.2f max %.2f" % (n, d.mean(), *np.percentile(d, [50, 90, 100])))
