"""Program run under `rocprofv3 --kernel-trace --pmc <one counter>` (profiles/scripts/pmc_passes.sh): every
hand-written kernel a few times at the shapes bench.py's `kernels` table uses, eager (no graphs)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rs_detection_amd import ops
from rs_detection_amd.utils import synthetic as syn
from rs_detection_amd.ops.dcn_v1 import deformable_col2im_gather_nhwc
from rs_detection_amd.ops.nms_rotated import _label_major_order
from rs_detection_amd.ops.box_coder import rotated_box_to_poly
dev = torch.device('cuda')
rng = np.random.default_rng(0)
a = torch.from_numpy(syn.s2anet_anchor_grid()).to(dev)
ks = [16, 100, 400, 40]
g = torch.from_numpy(np.concatenate([syn.dota_gt_boxes(rng, k) for k in ks])).to(dev)
ro = torch.tensor(np.concatenate([[0], np.cumsum(ks)]), dtype=torch.int32, device=dev)
out = torch.empty((sum(ks), a.shape[0]), device=dev)
lab = torch.ones(sum(ks), dtype=torch.int32, device=dev)
for _ in range(4):
    ops.box_iou_rotated_grouped(g, ro, max(ks), a, out=out)
    ops.assign_wrt_overlaps(out, ro, max(ks), 0.5, 0.4, 0.0, True, True, lab, 0)
B, C, H = 4, 256, 128
x = torch.randn(B, C, H, H, device=dev)
off = torch.randn(B, 18, H, H, device=dev)
xn = x.permute(0, 2, 3, 1).contiguous()
for _ in range(3):
    ops.deformable_im2col(x, off, (3, 3), (1, 1), (1, 1), (1, 1))
    colT = ops.deformable_im2col_nhwc(xn, off, (3, 3), (1, 1), (1, 1), (1, 1))
    deformable_col2im_gather_nhwc(colT, off, xn.shape, (3, 3), (1, 1), (1, 1), (1, 1))
del colT, x, xn
d, s, l = syn.nms_cluster_boxes(5344)
d6 = torch.from_numpy(np.concatenate([d, l[:, None].astype(np.float32)], 1)).to(dev)
sc, lb = torch.from_numpy(s).to(dev), torch.from_numpy(l).to(dev)
lorder = _label_major_order(sc, lb).int()
order = torch.argsort(sc, descending=True, stable=True).int()
for _ in range(3):
    ops.nms_rotated_keep_mask(d6, lorder, 0.1, 6, label_major=True)
    ops.nms_rotated_keep_mask(d6, order, 0.1, 6)
N, C, H, R = 2, 256, 256, 512
feat = torch.randn(N, C, H, H, device=dev, requires_grad=True)
b = syn.dota_gt_boxes(np.random.default_rng(3), R).astype(np.float32)
rois = torch.from_numpy(np.concatenate([np.random.default_rng(4).integers(0, N, (R, 1)).astype(np.float32), b], 1)).to(dev)
for _ in range(3):
    y = ops.roi_align_rotated_v1(feat, rois, (7, 7), 0.25, 2)
    torch.autograd.grad(y.sum(), feat)
del feat, y
N, C, H = 2, 256, 128
f = torch.randn(N, C, H, H, device=dev, requires_grad=True)
yc, xc = np.meshgrid(8.0 * np.arange(H), 8.0 * np.arange(H), indexing="ij")
r = np.random.default_rng(3)
bx = np.stack([xc[None] + 32 * r.standard_normal((N, H, H)), yc[None] + 32 * r.standard_normal((N, H, H)),
               32 * np.exp(r.standard_normal((N, H, H))), 32 * np.exp(r.standard_normal((N, H, H))),
               -np.pi / 2 * r.random((N, H, H))], -1).astype(np.float32)
bx = torch.from_numpy(bx).to(dev)
for _ in range(3):
    for pts in (1, 5):
        y = ops.feature_refine(f, bx, 0.125, pts)
        torch.autograd.grad(y.sum(), f)
p = torch.randn(20000, 24, 2, device=dev) * 20
m = torch.rand(20000, 24, device=dev) > 0.6
for _ in range(3):
    ops.convex_sort(p, m)
dd, ss, _ = syn.nms_cluster_boxes(2000)
dets = torch.cat([rotated_box_to_poly(torch.from_numpy(dd).to(dev)), torch.from_numpy(ss).to(dev)[:, None]], 1).contiguous()
for _ in range(2):
    ops.poly_nms(dets, 0.1)
torch.cuda.synchronize(); print("done")
