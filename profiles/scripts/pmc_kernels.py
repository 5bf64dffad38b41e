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
prep = ops.prepare_boxes(a, heavy_from=20480)
pgt = ops.prepare_boxes(g)
for _ in range(4):
    ops.box_iou_rotated_grouped(g, ro, max(ks), a, out=out)
    ops.assign_wrt_overlaps(out, ro, max(ks), 0.5, 0.4, 0.0, True, True, lab, 0)
    ops.box_iou_rotated_tiled(g, a, ro, ks=ks, out=out, prepared=prep, prepared1=pgt)
    ops.anchor_target_rotated(a, g, lab, ro, ks, 0.5, 0.4, 0.0, prepared=prep, prepared_gt=pgt, two_tier=False)
    ops.anchor_target_rotated(a, g, lab, ro, ks, 0.5, 0.4, 0.0, prepared=prep, prepared_gt=pgt, two_tier=True)
    ops.box_iou_rotated_fast(g, a, ro, ks=ks, out=out, prepared=prep)
    ops.prepare_boxes(a)
# algorithmic bytes per launch of the kernels this driver runs, keyed by a substring of the kernel name;
# kernels that share one figure (the launches of one C-ABI call) carry the same "call" tag and are summed by roofline.py
n1, A = sum(ks), a.shape[0]
iou_b = 20 * (n1 + A) + 4 * n1 * A
ALG = {
    "iou_prepare_kernel": dict(call="box_iou_rotated (3 launches)", bytes=iou_b),
    "iou_filter_kernel": dict(call="box_iou_rotated (3 launches)", bytes=iou_b),
    "iou_clip_kernel": dict(call="box_iou_rotated (3 launches)", bytes=iou_b),
    "iou_tile_kernel<0, 0>": dict(call="box_iou_rotated_tiled (1 launch)", bytes=iou_b),
    "assign_row_kernel": dict(call="assign_wrt_overlaps", bytes=2 * 4 * n1 * A + 12 * len(ks) * A),
    "assign_col_kernel": dict(call="assign_wrt_overlaps", bytes=2 * 4 * n1 * A + 12 * len(ks) * A),
    # fused anchor targets: boxes in, 56 B of targets per anchor out, no matrix
    "iou_tile_kernel<0, 1>": dict(call="anchor_target_rotated (2 launches)", bytes=20 * (n1 + A) + 56 * len(ks) * A),
    "at_finish_kernel": dict(call="anchor_target_rotated (2 launches)", bytes=20 * (n1 + A) + 56 * len(ks) * A),
    "at_prepare_kernel": dict(call="iou_prepare", bytes=(20 + 40) * A),
    # round 3: two-tier forms
    "iou_fast_tile_kernel": dict(call="box_iou_rotated_fast (1 launch)", bytes=iou_b),
    "at_tile2_kernel": dict(call="anchor_target_rotated two-tier (2 launches)", bytes=20 * (n1 + A) + 56 * len(ks) * A),
    "at_finish_kernel<1>": dict(call="anchor_target_rotated two-tier (2 launches)", bytes=20 * (n1 + A) + 56 * len(ks) * A),
}
ALG["at_finish_kernel<0>"] = ALG.pop("at_finish_kernel")
B, C, H = 4, 256, 128
ALG["deform_im2col_taps_kernel"] = dict(call="deform_im2col", bytes=4 * (C * H * H * B + 18 * H * H * B + 9 * C * H * H * B))
ALG["deform_im2col_nhwc_kernel"] = dict(call="deform_im2col_nhwc", bytes=4 * (C * H * H * B + 18 * H * H * B + 9 * C * H * H * B))
# implicit-GEMM AlignConv: SURVEY 8(d)'s fused-variant bytes (input + output + weights + offsets; no 9*C column term)
O_ = 256
ALG["alignconv_fwd_mfma_kernel<unsigned short"] = dict(call="alignconv_mfma_bf16", bytes=2 * (C * H * H * B + O_ * H * H * B + O_ * 9 * C) + 4 * 18 * H * H * B,
                                                       flops=2.0 * B * H * H * O_ * 9 * C)
ALG["alignconv_fwd_mfma_kernel<float"] = dict(call="alignconv_mfma_f32", bytes=4 * (C * H * H * B + O_ * H * H * B + O_ * 9 * C) + 4 * 18 * H * H * B,
                                              flops=2.0 * B * H * H * O_ * 9 * C)
for k in ("dcn_idx_count_kernel", "dcn_idx_scan_kernel", "dcn_idx_chunk_sum_kernel", "dcn_idx_fill_kernel", "dcn_gather_kernel"):
    ALG[k] = dict(call="deform_col2im (gather form, 5 launches)", bytes=4 * (C * H * H * B + 18 * H * H * B + 9 * C * H * H * B))
M = 5344
for k in ("nms_prepare_kernel", "nms_mask_kernel", "nms_sweep_kernel"):
    ALG[k] = dict(call="nms_rotated (3 launches)", bytes=4 * 6 * M + 2 * 8 * M * ((M + 63) // 64) + M)
ALG["rroi_forward_kernel"] = dict(call="rroi_align_v1 forward", bytes=4 * 512 * 256 * 49 * 17)
for k in ("rroi_idx_count_kernel", "rroi_idx_fill_kernel", "rroi_gather_kernel"):
    ALG[k] = dict(call="rroi_align_v1 backward (gather form)", bytes=4 * (512 * 49 * 256 + 2 * 256 * 256 * 256))
ALG["fr_forward_kernel"] = dict(call="feature_refine forward", bytes=4 * (2 * 2 * 256 * 128 * 128 + 5 * 2 * 128 * 128))
ALG["convex_sort_kernel"] = dict(call="convex_sort", bytes=20000 * (24 * 12 + 25 * 4))
import json
os.makedirs(os.environ.get("RSDET_ROOFLINE_DIR", "."), exist_ok=True)
json.dump(ALG, open(os.path.join(os.environ.get("RSDET_ROOFLINE_DIR", "."), "alg_bytes.json"), "w"), indent=1)
B, C, H = 4, 256, 128
x = torch.randn(B, C, H, H, device=dev)
off = torch.randn(B, 18, H, H, device=dev)
xn = x.permute(0, 2, 3, 1).contiguous()
for _ in range(3):
    ops.deformable_im2col(x, off, (3, 3), (1, 1), (1, 1), (1, 1))
    colT = ops.deformable_im2col_nhwc(xn, off, (3, 3), (1, 1), (1, 1), (1, 1))
    deformable_col2im_gather_nhwc(colT, off, xn.shape, (3, 3), (1, 1), (1, 1), (1, 1))
from rs_detection_amd import _lib as _L
_lib_ = _L.load()
_geom = _L.DcnGeom(C, H, H, 3, 3, 1, 1, 1, 1, 1, 1, B, 1)
for dt, entry in ((torch.bfloat16, _lib_.rsdet_alignconv_fwd_mfma_bf16), (torch.float32, _lib_.rsdet_alignconv_fwd_mfma_f32)):
    xq, wq = xn.to(dt), (torch.randn(256, 9 * C, device=dev) / 48).to(dt)
    oq = torch.empty((B, H, H, 256), dtype=dt, device=dev)
    for _ in range(3):
        entry(_L.ptr(xq), _L.ptr(off), _L.ptr(wq), _geom, 256, 1, _L.ptr(oq), None, _L.stream_ptr())
    del xq, wq, oq
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

# ---- round 3: the fused BatchNorm tails at the layer1 shape of the step, and the fused optimizer step
from rs_detection_amd.ops.bn_act import bn_act
Cb, Hb = 256, 256
bn = torch.nn.BatchNorm2d(Cb).to(dev).eval()
for tag, dt, cl in (("f32", torch.float32, False), ("bf16", torch.bfloat16, True)):
    xb = torch.randn(4, Cb, Hb, Hb, device=dev, dtype=dt)
    rb = torch.randn(4, Cb, Hb, Hb, device=dev, dtype=dt)
    if cl:
        xb, rb = xb.contiguous(memory_format=torch.channels_last), rb.contiguous(memory_format=torch.channels_last)
    xb.requires_grad_(True), rb.requires_grad_(True)
    for _ in range(3):
        yb = bn_act(xb, bn, rb, True)
        torch.autograd.grad(yb, (xb, rb, bn.weight, bn.bias), torch.randn_like(yb))
    del xb, rb, yb
es = {"f32": 4, "bf16": 2}
nb = 4 * Cb * Hb * Hb
ALG2 = {"bn_act_fwd_kernel<true, true, float>": dict(call="bn_act_forward_f32", bytes=3 * 4 * nb),
        "bn_act_bwd_kernel<true, float>": dict(call="bn_act_backward_f32", bytes=4 * 4 * nb),
        "bn_act_fwd_nhwc_kernel<true, true, unsigned short>": dict(call="bn_act_forward_bf16", bytes=3 * 2 * nb),
        "bn_act_bwd_nhwc_kernel<true, unsigned short": dict(call="bn_act_backward_bf16", bytes=4 * 2 * nb)}
from rs_detection_amd.optims.optimizer import FusedSGD
ps = [torch.nn.Parameter(torch.randn(36_000_000, device=dev))]
opt = FusedSGD(ps, lr=0.01, momentum=0.9, weight_decay=1e-4, grad_clip=dict(max_norm=35, norm_type=2))
for _ in range(3):
    ps[0].grad = torch.randn_like(ps[0])
    opt.step()
ALG2["mt_sqnorm_kernel"] = dict(call="fused SGD step (2 launches, 36 M fp32 parameters)", bytes=4 * 36_000_000 * (1 + 1 + 2 + 2))
ALG2["mt_sgd_kernel"] = dict(call="fused SGD step (2 launches, 36 M fp32 parameters)", bytes=4 * 36_000_000 * (1 + 1 + 2 + 2))
ALG.update(ALG2)
json.dump(ALG, open(os.path.join(os.environ.get("RSDET_ROOFLINE_DIR", "."), "alg_bytes.json"), "w"), indent=1)

# ---- round 3, second half: the channels_last fp32 tails, the 16-byte-lane bf16 tails, the pyramid canvas kernels and
# the orientation max-pool, at the shapes of the 4 x 1024^2 step
xb = torch.randn(4, Cb, Hb, Hb, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
rb = torch.randn(4, Cb, Hb, Hb, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
for _ in range(3):
    yb = bn_act(xb, bn, rb, True)
    torch.autograd.grad(yb, (xb, rb, bn.weight, bn.bias), torch.randn_like(yb))
del xb, rb, yb
ALG3 = {"bn_act_fwd_nhwc_kernel<true, true, float>": dict(call="bn_act_forward_nhwc_f32", bytes=3 * 4 * nb),
        "bn_act_bwd_nhwc_kernel<true, float": dict(call="bn_act_backward_nhwc_f32", bytes=4 * 4 * nb),
        "bn_act_fwd_nhwc8_kernel<true, true>": dict(call="bn_act_forward_bf16", bytes=3 * 2 * nb),
        "bn_act_bwd_nhwc8_kernel<true>": dict(call="bn_act_backward_bf16", bytes=4 * 2 * nb)}
from rs_detection_amd.ops.pyramid import canvas_layout, pyramid_pack, pyramid_unpack, canvas_bias_act
from rs_detection_amd.ops.orn import RotationInvariantPooling
sizes = [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)]
lay = canvas_layout(sizes, dev)
npx, ncv = sum(h * w for h, w in sizes), lay.Hc * lay.Wc
pool = RotationInvariantPooling(256, 8).to(dev)
for dt, es_ in ((torch.bfloat16, 2), (torch.float32, 4)):
    lv = [torch.randn(4, 256, h, w, device=dev).to(dt).contiguous(memory_format=torch.channels_last) for h, w in sizes]
    bias = torch.randn(256, device=dev)
    for _ in range(3):
        cv = pyramid_pack(lv, lay)
        pyramid_unpack(cv, lay, channels_last=True)
        yb = canvas_bias_act(cv, bias, lay, True)
        pb = pool(cv)
    small = pyramid_unpack(cv[:, :16].contiguous(memory_format=torch.channels_last), lay)     # 16-channel maps -> NCHW levels
    del lv, cv, yb, pb, small
tag = {2: "unsigned short", 4: "float"}
ALG3["pyramid_copy_nhwc16_kernel<true>"] = dict(call="pyramid_pack (256 ch, channels_last)", bytes=(2 + 4) * 4 * 256 * (npx + ncv) // 2)
ALG3["pyramid_copy_nhwc16_kernel<false>"] = dict(call="pyramid_unpack (256 ch, channels_last)", bytes=(2 + 4) * 4 * 256 * 2 * npx // 2)
ALG3["canvas_bias_act_nhwc8_kernel<true>"] = dict(call="canvas_bias_act bf16", bytes=2 * 2 * 4 * 256 * ncv)
ALG3["canvas_bias_act_nhwc_kernel<true, float>"] = dict(call="canvas_bias_act f32", bytes=2 * 4 * 4 * 256 * ncv)
ALG3["ori_maxpool8x4_kernel<unsigned short, false>"] = dict(call="ori_maxpool bf16", bytes=2 * 4 * ncv * (256 + 32))
ALG3["ori_maxpool8x4_kernel<float, false>"] = dict(call="ori_maxpool f32", bytes=4 * 4 * ncv * (256 + 32))
ALG.update(ALG3)
json.dump(ALG, open(os.path.join(os.environ.get("RSDET_ROOFLINE_DIR", "."), "alg_bytes.json"), "w"), indent=1)
torch.cuda.synchronize(); print("done (round 3 additions)")

# ---- round 4: the bf16 3x3 implicit GEMM of the head canvas (csrc/conv3x3_mfma.hip), plain and with the fused epilogue;
# the VAN elementwise tails (csrc/van_ops.hip) at a stage-1 and a stage-3 map; the AdamW step over 60 M parameters
Bc, Cc, Hc_, Wc_ = 4, 256, 128, 196
xq = torch.randn(Bc, Cc, Hc_, Wc_, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
wq = (torch.randn(Cc, Cc, 3, 3, device=dev) * 0.05).bfloat16().contiguous(memory_format=torch.channels_last)
bq = torch.randn(Cc, device=dev)
lq = (torch.rand(Hc_ * Wc_, device=dev) > 0.14).to(torch.uint8)
oq = torch.empty((Bc, Cc, Hc_, Wc_), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
for _ in range(4):
    _lib_.rsdet_conv3x3_fwd_mfma_bf16(_L.ptr(xq), _L.ptr(wq), None, None, Bc, Hc_, Wc_, Cc, Cc, 0, _L.ptr(oq), _L.stream_ptr())
    _lib_.rsdet_conv3x3_fwd_mfma_bf16(_L.ptr(xq), _L.ptr(wq), _L.ptr(bq), _L.ptr(lq), Bc, Hc_, Wc_, Cc, Cc, 1, _L.ptr(oq), _L.stream_ptr())
    torch.nn.functional.conv2d(xq, wq, None, 1, 1)          # the MIOpen / CK kernel beside it in the same trace
ALG4 = {"conv3x3_fwd_mfma_bf16_kernel": dict(call="conv3x3_mfma bf16 (head canvas 4x128x196x256)",
                                             bytes=2 * (2 * Bc * Hc_ * Wc_ * Cc + 9 * Cc * Cc),
                                             flops=2.0 * Bc * Hc_ * Wc_ * Cc * 9 * Cc)}
gq_ = torch.randn(Bc, Cc, Hc_, Wc_, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
gwq = torch.empty((Cc, Cc, 3, 3), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
nbw = _lib_.rsdet_conv3x3_wrw_mfma_ws_size(Bc, Hc_, Wc_, Cc, Cc)
wsw = torch.empty((nbw,), dtype=torch.uint8, device=dev)
# (our launches first, on rotating operand sets like bench.py's row -- the library's kernel beside them in the trace runs after)
gqs, xqs = [gq_] + [torch.randn_like(gq_) for _ in range(3)], [xq] + [torch.randn_like(xq) for _ in range(3)]
for i in range(8):
    _lib_.rsdet_conv3x3_wrw_mfma_bf16(_L.ptr(gqs[i & 3]), _L.ptr(xqs[i & 3]), Bc, Hc_, Wc_, Cc, Cc, _L.ptr(gwq), 1, _L.ptr(wsw), nbw, _L.stream_ptr())
for _ in range(4):
    torch.ops.aten.convolution_backward(gq_, xq, wq, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False))
del gqs, xqs
for kname in ("conv3x3_wrw_mfma_bf16_kernel", "conv3x3_wrw_fold_kernel"):
    ALG4[kname] = dict(call="conv3x3_wrw_mfma bf16 (head canvas 4x128x196x256, 2 launches)",
                       bytes=2 * (2 * Bc * Hc_ * Wc_ * Cc + 9 * Cc * Cc), flops=2.0 * Bc * Hc_ * Wc_ * Cc * 9 * Cc)
del xq, oq, gq_, wsw
from rs_detection_amd.ops import van_fused
for shp in ((2, 64, 256, 256), (2, 320, 64, 64)):
    t1, t2, t3 = (torch.randn(shp, device=dev, requires_grad=True) for _ in range(3))
    cb, cs = torch.randn(shp[1], device=dev, requires_grad=True), torch.rand(shp[1], device=dev, requires_grad=True)
    for _ in range(3):
        van_fused.bias_gelu(t1, cb).sum().backward()
        van_fused.gate(t1, t2, cb).sum().backward()
        van_fused.residual(t1, t2, cb, t3, cs).sum().backward()
n_el = 2 * 64 * 256 * 256 + 2 * 320 * 64 * 64
ALG4["van_bias_gelu_fwd_kernel"] = dict(call="van bias_gelu fwd", bytes=4 * 2 * n_el // 2)
ALG4["van_bias_gelu_bwd_kernel"] = dict(call="van bias_gelu bwd", bytes=4 * 3 * n_el // 2)
ALG4["van_gate_fwd_kernel"] = dict(call="van gate fwd", bytes=4 * 3 * n_el // 2)
ALG4["van_gate_bwd_kernel"] = dict(call="van gate bwd", bytes=4 * 5 * n_el // 2)
ALG4["van_residual_fwd_kernel"] = dict(call="van residual fwd", bytes=4 * 4 * n_el // 2)
ALG4["van_residual_bwd_kernel"] = dict(call="van residual bwd", bytes=4 * 4 * n_el // 2)
from rs_detection_amd.optims.optimizer import FusedAdamW
pp = [torch.nn.Parameter(torch.randn(15_000_000, device=dev)) for _ in range(4)]
opt = FusedAdamW(pp, lr=1e-4, weight_decay=0.05, grad_clip=dict(max_norm=35, norm_type=2))
for _ in range(3):
    for p_ in pp:
        p_.grad = torch.randn_like(p_)
    opt.step()
ALG4["mt_adamw_kernel"] = dict(call="FusedAdamW update, 60 M parameters (the norm launch is the SGD row's)", bytes=4 * 60_000_000 * (4 + 3))
ALG.update(ALG4)
torch.cuda.synchronize(); print("done (round 4 additions)")

# ---- round 5: the streaming 1x1 GEMM of the bf16 trunk (csrc/gemm1x1_mfma.hip) forward and the two backward-data modes at
# the layer2 shapes of the step; the tower's gate-fused backward-data (conv3x3_fwd_mfma<true>); FeatureRefine on
# channels-last maps.  Same calls as bench.gemm1x1_rows / next_row_kernels.
import bench as _bench
_bench.gemm1x1_rows(dev, 4)
M5, C0, C1 = 4 * 128 * 128, 512, 128
ALG5 = {"gemm1x1_bn_act_mfma_bf16_kernel<4, 1, 3>": dict(call="conv1x1 + bn + identity + relu forward (65536 x 128 -> 512)",
                                                         bytes=2 * (M5 * C1 + C0 * C1 + 2 * M5 * C0), flops=2.0 * M5 * C0 * C1),
        "gemm1x1_bn_act_mfma_bf16_kernel<2, 2, 3>": dict(call="conv1x1 backward-data + bn backward in the epilogue (65536 x 512 -> 128)",
                                                         bytes=2 * (M5 * C0 + C0 * C1 + 2 * M5 * C1), flops=2.0 * M5 * C0 * C1),
        "gemm1x1_bn_act_mfma_bf16_kernel<4, 3, 3>": dict(call="conv1x1 backward-data + identity gradient (65536 x 128 -> 512)",
                                                         bytes=2 * (M5 * C1 + C0 * C1 + 2 * M5 * C0), flops=2.0 * M5 * C0 * C1)}
xq = torch.randn(Bc, Cc, Hc_, Wc_, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
c1q = torch.randn(Bc, Cc, Hc_, Wc_, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
oq = torch.empty_like(xq)
gbq = torch.empty(Cc, device=dev)
nbg = _lib_.rsdet_conv3x3_dgrad_gate_ws_size(Bc, Hc_, Wc_, Cc)
wsg = torch.empty((nbg,), dtype=torch.uint8, device=dev)
for _ in range(4):
    _lib_.rsdet_conv3x3_dgrad_gate_mfma_bf16(_L.ptr(xq), _L.ptr(wq), _L.ptr(c1q), Bc, Hc_, Wc_, Cc, Cc, _L.ptr(oq), _L.ptr(gbq),
                                             _L.ptr(wsg), nbg, _L.stream_ptr())
ALG5["conv3x3_fwd_mfma_bf16_kernel<true>"] = dict(call="conv3x3_mfma bf16 backward-data + the previous layer's ReLU gate (head canvas)",
                                                  bytes=2 * (3 * Bc * Hc_ * Wc_ * Cc + 9 * Cc * Cc),
                                                  flops=2.0 * Bc * Hc_ * Wc_ * Cc * 9 * Cc)
del xq, c1q, oq
Nf, Cf, Hf = 2, 256, 128
ff = torch.randn(Nf, Cf, Hf, Hf, device=dev).contiguous(memory_format=torch.channels_last)
bxf = torch.rand(Nf, Hf, Hf, 5, device=dev) * 100
from rs_detection_amd import ops as _ops
for _ in range(3):
    _ops.feature_refine(ff, bxf, 0.125, 5)
    _ops.feature_refine(ff, bxf, 0.125, 1)
ALG5["fr_forward_nhwc_kernel<5>"] = dict(call="fr_forward_nhwc<5>", bytes=4 * (2 * Nf * Cf * Hf * Hf + 5 * Nf * Hf * Hf))
ALG5["fr_forward_nhwc_kernel<1>"] = dict(call="fr_forward_nhwc<1>", bytes=4 * (2 * Nf * Cf * Hf * Hf + 5 * Nf * Hf * Hf))
# ---- round 6: the fp32 MFMA GEMMs / weight gradients of the VAN block (csrc/van_gemm.hip) at the stage-3 shapes of the
# Oriented R-CNN step (2 images, 64 x 64, C = 320, R = 1280), the depthwise weight gradients, the radix-select sampler and
# the batched proposal kernels (csrc/orpn.hip)
Nv, Cv, Rv, Pv = 2, 320, 1280, 64 * 64
xv = torch.randn(Nv, Cv, Pv, device=dev)
hv = torch.randn(Nv, Rv, Pv, device=dev)
wcr = torch.randn(Rv, Cv, device=dev) * 0.05
wrc = torch.randn(Cv, Rv, device=dev) * 0.05
wcc = torch.randn(Cv, Cv, device=dev) * 0.05
vv = [torch.rand(Rv, device=dev) for _ in range(4)]
ov = torch.empty(Nv, Rv, Pv, device=dev)
oc, oc1 = torch.empty(Nv, Cv, Pv, device=dev), torch.empty(Nv, Cv, Pv, device=dev)
S_cr, S_cc = _lib_.rsdet_van_wgrad_f32_splits(Cv, Rv, Pv, Nv), _lib_.rsdet_van_wgrad_f32_splits(Cv, Cv, Pv, Nv)
part = torch.empty(max(S_cr * Cv * Rv, S_cc * Cv * Cv), device=dev)
P_ = _L.ptr
for _ in range(4):
    # fc1 (bias), fc2 (affine: layer scale + shortcut), proj_1 (bias + GELU, two outputs), the MLP's backward-data with GELU'
    _lib_.rsdet_van_gemm_f32(P_(wcr), P_(xv), Rv, Cv, Pv, Nv, 1, P_(vv[0]), None, None, None, None, None, P_(ov), None, _L.stream_ptr())
    _lib_.rsdet_van_gemm_f32(P_(wrc), P_(hv), Cv, Rv, Pv, Nv, 4, None, P_(vv[1]), P_(vv[2]), None, P_(xv), None, P_(oc), None, _L.stream_ptr())
    _lib_.rsdet_van_gemm_f32(P_(wcc), P_(xv), Cv, Cv, Pv, Nv, 2, P_(vv[0]), None, None, None, None, None, P_(oc), P_(oc1), _L.stream_ptr())
    _lib_.rsdet_van_gemm_f32(P_(wcr), P_(xv), Rv, Cv, Pv, Nv, 6, None, None, None, None, P_(hv), None, P_(ov), None, _L.stream_ptr())
    _lib_.rsdet_van_wgrad_f32(P_(xv), P_(hv), Cv, Rv, Pv, Nv, P_(part), _L.stream_ptr())
    _lib_.rsdet_van_wgrad_f32(P_(xv), P_(oc), Cv, Cv, Pv, Nv, P_(part), _L.stream_ptr())
f_cr, f_cc = 2.0 * Nv * Pv * Cv * Rv, 2.0 * Nv * Pv * Cv * Cv
ALG6 = {
    "van_gemm_f32_kernel<5, 2, 1": dict(call="van_gemm fc1 1280 x 320 x 8192 (bias)", bytes=4 * (Nv * Pv * (Cv + Rv) + Cv * Rv), flops=f_cr),
    "van_gemm_f32_kernel<5, 2, 4": dict(call="van_gemm fc2 320 x 1280 x 8192 (layer scale + shortcut)", bytes=4 * (Nv * Pv * (Rv + 2 * Cv) + Cv * Rv), flops=f_cr),
    "van_gemm_f32_kernel<5, 1, 2": dict(call="van_gemm proj_1 320 x 320 x 8192 (bias + GELU, two outputs)", bytes=4 * (Nv * Pv * 3 * Cv + Cv * Cv), flops=f_cc),
    "van_gemm_f32_kernel<5, 2, 6": dict(call="van_gemm fc2 backward-data 1280 x 320 x 8192 (x GELU')", bytes=4 * (Nv * Pv * (Cv + 2 * Rv) + Cv * Rv), flops=f_cr),
    "van_wgrad_f32_kernel<5>": dict(call="van_wgrad 320 x 1280 and 320 x 320 over 8192 pixels (split-K partials)",
                                    bytes=4 * (Nv * Pv * (2 * Cv + Rv + Cv) + S_cr * Cv * Rv + S_cc * Cv * Cv), flops=f_cr + f_cc),
}
for (Cd, Hd, K, D) in ((1280, 64, 3, 1), (320, 64, 5, 1), (320, 64, 7, 3)):
    xd, god = torch.randn(2, Cd, Hd, Hd, device=dev), torch.randn(2, Cd, Hd, Hd, device=dev)
    gwd, gbd = torch.empty(Cd, 1, K, K, device=dev), torch.empty(Cd, device=dev)
    wsb = _lib_.rsdet_dwconv2d_backward_weight_ws_size(2, Cd, Hd, Hd, K)
    wsd = torch.empty((max(wsb, 4),), dtype=torch.uint8, device=dev)
    for _ in range(4):
        _lib_.rsdet_dwconv2d_backward_weight_f32(P_(god), P_(xd), None, 2, Cd, Hd, Hd, K, D, P_(gwd), P_(gbd), P_(wsd), wsb, _L.stream_ptr())
    key = "dwconv_wgrad_tile_kernel<7, 3>" if K == 7 else "dwconv_wgrad_kernel<%d, 1>" % K
    ALG6[key] = dict(call="dwconv backward-weight<%d,%d> 2 x %d x 64 x 64" % (K, D, Cd), bytes=4 * 2 * 2 * Cd * Hd * Hd)
from rs_detection_amd.ops import orpn as _orpn
n_a = 611072
gti = torch.from_numpy(np.where(rng.random(n_a) < 0.002, 1, np.where(rng.random(n_a) < 0.9, 0, -1)).astype(np.int32)).to(dev)
pri = torch.rand(n_a, device=dev)
for _ in range(4):
    _orpn.sample_masked(gti, None, 0, pri, 256, 128, -1.0)
for kname in ("sel_hist_kernel<unsigned int, 2", "sel_emit_kernel<unsigned int, 2", "sampler_final_kernel"):
    ALG6[kname] = dict(call="sample_masked: 256 of 611 072 anchors (3 counting passes + emit + final)", bytes=4 * 8 * n_a)
# matrix-free horizontal assignment at the Oriented RPN's size: level-major grid anchors (7 per position), K = 100
_anc = []
for _s in (4, 8, 16, 32, 64):
    _n = 1024 // _s
    _cy, _cx = np.meshgrid(np.arange(_n) * _s, np.arange(_n) * _s, indexing="ij")
    _c = np.stack([_cx, _cy], -1).reshape(-1, 1, 2).astype(np.float32)
    _wh = np.array([[_s * 8 * np.sqrt(r), _s * 8 / np.sqrt(r)] for r in (0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 4.0)], np.float32)
    _anc.append(np.concatenate([_c - _wh[None] / 2, _c + _wh[None] / 2], -1).reshape(-1, 4))
_anchors = torch.from_numpy(np.concatenate(_anc)).to(dev)
_ctr = rng.uniform(0, 1024, (100, 2)); _gwh = np.exp(rng.uniform(np.log(10), np.log(200), (100, 2)))
_gts = torch.from_numpy(np.concatenate([_ctr - _gwh / 2, _ctr + _gwh / 2], 1).astype(np.float32)).to(dev)
for _ in range(4):
    _orpn.hbb_assign(_anchors, _gts, 0.7, 0.3, 0.3, True, True)
for kname in ("hba_rowmax_kernel", "hba_col_kernel"):
    ALG6[kname] = dict(call="hbb_assign: %d anchors x K = 100 (row maxima + columns, no matrix)" % _anchors.shape[0],
                       bytes=(16 + 8) * _anchors.shape[0])
ALG5.update(ALG6)
# (the plain conv3x3 key of round 4 is a substring of the gated kernel's name: give the more specific key precedence)
ALG = dict(list(ALG5.items()) + [(k, v) for k, v in ALG.items()])
json.dump(ALG, open(os.path.join(os.environ.get("RSDET_ROOFLINE_DIR", "."), "alg_bytes.json"), "w"), indent=1)
torch.cuda.synchronize(); print("done (round 5 + 6 additions)")
