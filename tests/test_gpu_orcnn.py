"""GPU: Oriented-RCNN pieces (a19/a20): hbb NMS kernel, RoI extractor, one train step + eval of the full model."""
import os

import numpy as np
import oracle
import pytest
import torch

from conftest import dota_boxes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _np_hbb_nms(dets, thr, one=1.0):
    order = np.argsort(-dets[:, 4], kind="stable")
    keep, dead = [], np.zeros(len(dets), bool)
    area = (dets[:, 2] - dets[:, 0] + one) * (dets[:, 3] - dets[:, 1] + one)
    for a, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(i)
        for j in order[a + 1:]:
            if dead[j]:
                continue
            w = max(0.0, min(dets[i, 2], dets[j, 2]) - max(dets[i, 0], dets[j, 0]) + one)
            h = max(0.0, min(dets[i, 3], dets[j, 3]) - max(dets[i, 1], dets[j, 1]) + one)
            inter = np.float32(w) * np.float32(h)
            if inter / (area[i] + area[j] - inter) > thr:
                dead[j] = True
    return np.array(keep)


@pytest.mark.parametrize("n,thr", [(1, 0.5), (65, 0.5), (700, 0.8), (3000, 0.3)])
def test_hbb_nms_vs_numpy(cuda, n, thr):
    from rs_detection_amd.ops import nms
    rng = np.random.default_rng(n)
    c = rng.uniform(0, 300, (n, 2)).astype(np.float32)
    wh = rng.uniform(5, 80, (n, 2)).astype(np.float32)
    dets = np.concatenate([c - wh / 2, c + wh / 2, rng.uniform(0, 1, (n, 1)).astype(np.float32)], 1).astype(np.float32)
    got = nms(torch.from_numpy(dets).to(cuda), thr).cpu().numpy()
    assert (got == _np_hbb_nms(dets, thr)).all()
    assert nms(torch.zeros((0, 5), device=cuda), thr).numel() == 0


def test_roi_extractor_equals_per_level_gather(cuda):
    """Sync-free all-levels form == the reference's per-level boolean gather (oriented_single_level.py:104-112)."""
    from rs_detection_amd.models.roi_extractors.oriented_single_level import OrientedSingleRoIExtractor
    torch.manual_seed(0)
    ex = OrientedSingleRoIExtractor(dict(type='ROIAlignRotated_v1', output_size=7, sampling_ratio=2), 8,
                                    [4, 8, 16, 32], extend_factor=(1.4, 1.2)).to(cuda)
    feats = [torch.randn(2, 8, 256 // s, 256 // s, device=cuda) for s in (4, 8, 16, 32)]
    rng = np.random.default_rng(3)
    b = dota_boxes(rng, 60, 256, 8, 250, 200)
    rois = torch.from_numpy(np.concatenate([rng.integers(0, 2, (60, 1)).astype(np.float32), b], 1)).to(cuda)
    got = ex(feats, rois)
    r = ex.roi_rescale(rois, ex.extend_factor)
    lv = ex.map_roi_levels(r, 4)
    want = torch.zeros_like(got)
    for i in range(4):
        m = lv == i
        if m.any():
            want[m] = ex.roi_layers[i](feats[i], r[m])
    torch.testing.assert_close(got, want, atol=1e-5, rtol=1e-5)
    assert len(set(lv.tolist())) >= 3
    # independent known answer: NumPy restatement of the RoI extension (roi_extractors/oriented_single_level.py:85-88:
    # (h_f, w_f) = extend_factor, w *= w_f, h *= h_f) and of the level map (:68-70: floor(log2(sqrt(w*h) / 56 + 1e-6)),
    # clamped) + the ORACLE's RROIAlign on the mapped level
    ext_r = rois.cpu().numpy().astype(np.float64)
    ext_r[:, 3] *= 1.2
    ext_r[:, 4] *= 1.4
    lv_np = np.clip(np.floor(np.log2(np.sqrt(ext_r[:, 3] * ext_r[:, 4]) / 56 + 1e-6)), 0, 3).astype(np.int64)
    assert (lv_np == lv.cpu().numpy()).all()
    want_np = np.zeros(tuple(got.shape), np.float32)
    for i, s_ in enumerate((4, 8, 16, 32)):
        m = lv_np == i
        if m.any():
            want_np[m] = oracle.c().rroi_align_v1_forward(feats[i].cpu().numpy(), ext_r[m].astype(np.float32), (7, 7),
                                                          1.0 / s_, 2)
    assert np.abs(got.cpu().numpy() - want_np).max() <= 1e-4
    # the one-launch forward (every RoI on the map of its own level) == the per-level launches over all RoIs, added up:
    # values bit for bit (x + 0 is x), feature gradients to rounding (same per-level backward on both sides)
    assert ex._one_launch and ex._levels_variant == "v1"
    fa = [f.clone().requires_grad_() for f in feats]
    fb = [f.clone().requires_grad_() for f in feats]
    ya = ex(fa, rois)
    ex._one_launch = False
    try:
        yb = ex(fb, rois)
    finally:
        ex._one_launch = True
    assert torch.equal(ya, got) and torch.equal(ya, yb)
    go = torch.randn_like(ya)
    ga, gb = torch.autograd.grad(ya, fa, go, retain_graph=True), torch.autograd.grad(yb, fb, go)
    for a, b_ in zip(ga, gb):
        assert float((a - b_).abs().max()) <= 1e-5 * max(float(b_.abs().max()), 1e-6)
    # ... and the one-index backward (default) == the per-level backward behind the same one-launch forward
    import importlib
    rr = importlib.import_module("rs_detection_amd.ops.roi_align_rotated_v1")   # (ops re-exports a function of this name)
    assert rr._LEVELS_BACKWARD
    rr._LEVELS_BACKWARD = False
    try:
        gc = torch.autograd.grad(ya, fa, go, retain_graph=True)
    finally:
        rr._LEVELS_BACKWARD = True
    for a, c_ in zip(ga, gc):
        assert float((a - c_).abs().max()) <= 1e-5 * max(float(c_.abs().max()), 1e-6)
    g2 = torch.autograd.grad(ya, [fa[1], fa[3]], go)         # only some levels want a gradient
    assert float((g2[0] - ga[1]).abs().max()) <= 1e-5 * float(ga[1].abs().max())
    assert float((g2[1] - ga[3]).abs().max()) <= 1e-5 * max(float(ga[3].abs().max()), 1e-6)
    with torch.autocast("cuda", dtype=torch.bfloat16):     # bf16 maps of an autocast step: widened, fp32 result
        yh = ex([f.bfloat16() for f in feats], rois)
    assert yh.dtype == torch.float32
    assert float((yh - ex([f.bfloat16().float() for f in feats], rois)).abs().max()) == 0.0


def test_oriented_rcnn_train_step_and_eval(cuda):
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.config import Config
    from rs_detection_amd.utils.registry import MODELS, build_from_cfg
    from rs_detection_amd.utils.general import parse_losses
    from rs_detection_amd.utils import synthetic as syn
    cfg = Config(os.path.join(ROOT, "configs", "orcnn", "orcnn_van3_7_anchor.py")).dump()["model"]
    cfg["backbone"] = dict(type="van_b0", img_size=256, num_stages=4, out_indices=(0, 1, 2, 3))
    cfg["neck"]["in_channels"] = [32, 64, 160, 256]
    torch.manual_seed(0)
    model = build_from_cfg(cfg, MODELS).to(cuda)
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    images = torch.randn(2, 3, 256, 256, device=cuda)
    targets = []
    for t in syn.synthetic_targets(2, img=256, num_classes=10):
        t = dict(t)
        t["rboxes"] = torch.from_numpy(t["rboxes"][:12]).to(cuda)
        t["labels"] = torch.from_numpy(t["labels"][:12]).to(cuda)
        t["hboxes"] = None
        targets.append(t)
    model.train()
    losses = model(images, targets)
    assert set(losses) == {"loss_cls", "orcnn_bbox_loss", "loss_rpn_cls", "loss_rpn_bbox"}
    total, parsed = parse_losses(losses)
    assert torch.isfinite(total)
    total.backward()
    opt.step()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    assert model.bbox_head.fc_reg.weight.grad.abs().sum() > 0 and model.rpn.rpn_reg.weight.grad.abs().sum() > 0
    assert model.backbone.patch_embed1.proj.weight.grad.abs().sum() > 0  # gradient flows through RROIAlign
    model.eval()
    with torch.no_grad():
        res = model(images, targets)
    assert len(res) == 2
    polys, scores, labels = res[0]
    assert polys.shape[1] == 8 and polys.shape[0] == scores.shape[0] == labels.shape[0]


def test_train_step_on_fixed_size_samples_equals_the_index_list_form_and_never_synchronises(cuda, monkeypatch):
    """The train step of Oriented R-CNN runs on FIXED-SIZE proposal lists and samples (OrientedRPNHead.masked:
    sampler.sample_masked, _nms_fixed, OrientedHead._forward_train_masked) so that nothing brings a count to the host.
    Given the same draws it is the reference-shaped step (index lists of data-dependent length, oriented_rpn_head.py:
    274-366, oriented_head.py:566-588) loss for loss; and under torch's sync debug mode it raises nowhere."""
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.config import Config
    from rs_detection_amd.models.boxes.sampler import RandomSampler
    from rs_detection_amd.utils.registry import MODELS, build_from_cfg
    from rs_detection_amd.utils import synthetic as syn
    cfg = Config(os.path.join(ROOT, "configs", "orcnn", "orcnn_van3_7_anchor.py")).dump()["model"]
    cfg["backbone"] = dict(type="van_b0", img_size=256, num_stages=4, out_indices=(0, 1, 2, 3))
    cfg["neck"]["in_channels"] = [32, 64, 160, 256]
    torch.manual_seed(0)
    model = build_from_cfg(cfg, MODELS).to(cuda).train()
    images = torch.randn(2, 3, 256, 256, device=cuda)
    targets = []
    for t in syn.synthetic_targets(2, img=256, num_classes=10):
        t = dict(t)
        t["rboxes"] = torch.from_numpy(t["rboxes"][:14]).to(cuda)
        t["labels"] = torch.from_numpy(t["labels"][:14]).to(cuda)
        t["hboxes"] = None
        targets.append(t)
    pri = torch.from_numpy(np.random.default_rng(5).random(1 << 20)).to(cuda)          # float64: no ties
    monkeypatch.setattr(RandomSampler, "priorities", staticmethod(lambda n, dev: pri[:n]))
    monkeypatch.setattr(RandomSampler, "random_choice",
                        staticmethod(lambda gallery, num: gallery[torch.topk(pri[gallery], num)[1]]))
    for m in model.modules():                    # same batch statistics in both passes
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.momentum = 0.0
    assert model.rpn.masked
    fixed = model(images, targets)
    model.rpn.masked = False
    lists = model(images, targets)
    model.rpn.masked = True
    for k in lists:
        a = torch.stack(list(fixed[k])) if isinstance(fixed[k], (list, tuple)) else fixed[k]
        b = torch.stack(list(lists[k])) if isinstance(lists[k], (list, tuple)) else lists[k]
        assert torch.allclose(a.float().reshape(-1), b.float().reshape(-1), rtol=2e-4, atol=1e-6), (k, a, b)
    # proposals: the fixed-size list holds the variable-length list's rows, in order, then zero rows
    feats = model.neck(model.backbone(images))
    with torch.no_grad():
        outs = [list(o) for o in zip(*[model.rpn.forward_single(f) for f in feats])]
        fx = model.rpn.get_bboxes(*outs, targets, fixed=True)
        vl = model.rpn.get_bboxes(*outs, targets, fixed=False)
    for (d, real), v in zip(fx, vl):
        n = v.shape[0]
        assert d.shape[0] == model.rpn.nms_post and int(real.sum()) == n and bool(real[:n].all())
        assert torch.equal(d[:n], v) and not bool(d[n:].any())
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        from rs_detection_amd.utils.general import parse_losses
        total, _ = parse_losses(model(images, targets))
        total.backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert bool(torch.isfinite(total))


def test_batched_inference_equals_per_image_inference(cuda):
    """Every image pools its RoI features from ITS OWN pyramid slice (the reference hands the whole batch to a
    per-image call whose RoIs all carry batch index 0, oriented_head.py:610-613: images i>0 then read image 0)."""
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.config import Config
    from rs_detection_amd.utils.registry import MODELS, build_from_cfg
    from rs_detection_amd.utils import synthetic as syn
    cfg = Config(os.path.join(ROOT, "configs", "orcnn", "orcnn_van3_7_anchor.py")).dump()["model"]
    cfg["backbone"] = dict(type="van_b0", img_size=256, num_stages=4, out_indices=(0, 1, 2, 3))
    cfg["neck"]["in_channels"] = [32, 64, 160, 256]
    cfg["bbox_head"]["score_thresh"] = 0.0          # random weights: keep detections to compare
    torch.manual_seed(0)
    model = build_from_cfg(cfg, MODELS).to(cuda).eval()
    images = torch.randn(2, 3, 256, 256, device=cuda)
    images[1] += 2.0 * torch.randn(3, 1, 1, device=cuda)          # clearly different second image
    targets = [dict(t, hboxes=None) for t in syn.synthetic_targets(2, img=256, num_classes=10)]
    with torch.no_grad():
        both = model(images, targets)
        single = [model(images[i:i + 1], targets[i:i + 1])[0] for i in range(2)]
    def top(res, k=30):
        p, s, l = res
        o = torch.argsort(s, descending=True, stable=True)[:k]
        return p[o], s[o], l[o]

    def dist(a, b):
        """mean score / corner distance of the k best detections (MIOpen may pick another algorithm for another batch
        size, so memberships near the NMS / top-k borders can flip: compare the confident ones, with a tolerance)"""
        (pa, sa, _), (pb, sb, _) = top(a), top(b)
        k = min(len(sa), len(sb))
        return float((sa[:k] - sb[:k]).abs().mean()), float((pa[:k] - pb[:k]).abs().mean()), k

    for i in range(2):
        ds, dp, k = dist(both[i], single[i])
        assert k >= 10 and ds < 2e-3 and dp < 1.0, (i, ds, dp, k)
    # the bug this guards against: image 1 of the batch pooled its RoI features from image 0
    ds_wrong, dp_wrong, _ = dist(both[1], single[0])
    ds_right, dp_right, _ = dist(both[1], single[1])
    assert dp_right < 0.2 * dp_wrong or ds_right < 0.2 * ds_wrong, (ds_right, dp_right, ds_wrong, dp_wrong)


def test_config3_van_b3_1024_train_step_and_eval(cuda):
    """BASELINE.json configs[3] AS CONFIGURED: configs/orcnn/orcnn_van3_7_anchor.py unmodified (VAN-B3 trunk, 7 anchor
    ratios, 2 000 proposals -> 512 sampled RoIs) on one 1024 x 1024 tile: one train step (finite losses, gradient
    through RROIAlign into the trunk, RoI count and level histogram equal to a NumPy count) and one eval pass.
    The trunk is random-initialised (no ImageNet weights offline: `pretrained=True` warns and stays random)."""
    import warnings
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.config import Config
    from rs_detection_amd.utils.registry import MODELS, build_from_cfg
    from rs_detection_amd.utils.general import parse_losses
    from rs_detection_amd.utils import synthetic as syn
    cfg = Config(os.path.join(ROOT, "configs", "orcnn", "orcnn_van3_7_anchor.py")).dump()["model"]
    assert cfg["backbone"]["type"] == "van_b3" and cfg["backbone"]["img_size"] == 1024
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = build_from_cfg(cfg, MODELS).to(cuda)
    n_par = sum(p.numel() for p in model.backbone.parameters())
    assert 40e6 < n_par < 50e6, n_par                                    # VAN-B3: 44.8 M parameters (van.py, b3 widths)
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.05)
    images = torch.randn(1, 3, 1024, 1024, device=cuda)
    t = dict(syn.synthetic_targets(1, it=2, img=1024, num_classes=10)[0])
    K = len(t["rboxes"])
    assert K >= 20
    t["rboxes"], t["labels"], t["hboxes"] = torch.from_numpy(t["rboxes"]).to(cuda), torch.from_numpy(t["labels"]).to(cuda), None
    seen = {}

    def grab(mod, args):
        seen["rois"] = args[1].detach().cpu().numpy()
        seen["n_feats"] = len(args[0])
    h = model.bbox_head.bbox_roi_extractor.register_forward_pre_hook(grab)
    props = {}
    def count(mod, args, out):
        # training: (fixed-size proposals, mask of the real ones) per image -- count the real ones
        props["n"] = [int(p[1].sum()) if isinstance(p, tuple) else len(p) for p in out[0]]
        # (a hook that returns a value would replace the output)
    hp = model.rpn.register_forward_hook(count)
    model.train()
    losses = model(images, [t])
    h.remove(), hp.remove()
    assert set(losses) == {"loss_cls", "orcnn_bbox_loss", "loss_rpn_cls", "loss_rpn_bbox"}
    assert len(losses["loss_rpn_cls"]) == 5
    total, _ = parse_losses(losses)
    assert torch.isfinite(total)
    total.backward()
    opt.step()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    assert model.bbox_head.fc_reg.weight.grad.abs().sum() > 0 and model.rpn.rpn_reg.weight.grad.abs().sum() > 0
    assert model.backbone.patch_embed1.proj.weight.grad.abs().sum() > 0   # through RROIAlign and FPN into the trunk
    # 2 000 proposals (nms_post) -> K gts prepended -> 512 RoIs of image 0, all four RoI levels in play
    assert props["n"] == [2000], props
    rois = seen["rois"]
    assert rois.shape == (512, 6) and (rois[:, 0] == 0).all() and seen["n_feats"] == 4
    ext = rois.astype(np.float64)                                          # oriented_single_level.py:85-88, :68-70
    lv = np.clip(np.floor(np.log2(np.sqrt((ext[:, 3] * 1.2) * (ext[:, 4] * 1.4)) / 56 + 1e-6)), 0, 3).astype(np.int64)
    ex = model.bbox_head.bbox_roi_extractor
    got_lv = ex.map_roi_levels(ex.roi_rescale(torch.from_numpy(rois).to(cuda), ex.extend_factor), 4).cpu().numpy()
    assert (np.bincount(got_lv, minlength=4) == np.bincount(lv, minlength=4)).all()
    assert (np.bincount(lv, minlength=4) > 0).sum() >= 2, np.bincount(lv, minlength=4)
    model.eval()
    with torch.no_grad():
        res = model(images, [dict(t, img_size=(1024, 1024), scale_factor=1.0)])
    polys, scores, labels = res[0]
    assert polys.shape[1] == 8 and polys.shape[0] == scores.shape[0] == labels.shape[0]
    assert torch.isfinite(polys).all() and torch.isfinite(scores).all()


@pytest.mark.parametrize("K,A,mode", [(37, 5000, "iou"), (400, 20011, "iou"), (1, 3, "iof"), (64, 257, "iof")])
def test_hbb_overlaps_kernel_is_bit_identical_to_the_tensor_form(cuda, K, A, mode):
    """models/boxes/iou_calculator.bbox_overlaps on the GPU (csrc/assign.hip: bbox_overlaps_kernel, one pass) against the
    reference's tensor expression (iou_calculator.py:164-257) evaluated by torch on the same device: same operations in
    the same order -> every element equal, degenerate and disjoint boxes included; 5-column inputs (a score column) are
    read through their row stride."""
    from rs_detection_amd.models.boxes.iou_calculator import BboxOverlaps2D
    g = torch.Generator().manual_seed(K + A)
    def boxes(n):
        xy = torch.rand((n, 2), generator=g) * 900
        wh = torch.rand((n, 2), generator=g) * 200
        b = torch.cat([xy, xy + wh, torch.rand((n, 1), generator=g)], 1)
        b[::7, 2:4] = b[::7, :2]                              # zero-area boxes
        return b.to(cuda)
    b1, b2 = boxes(K), boxes(A)
    got = BboxOverlaps2D()(b1, b2, mode)
    x1, x2 = b1[:, :4], b2[:, :4]
    a1 = (x1[:, 2] - x1[:, 0]) * (x1[:, 3] - x1[:, 1])
    a2 = (x2[:, 2] - x2[:, 0]) * (x2[:, 3] - x2[:, 1])
    lt = torch.max(x1[:, None, :2], x2[None, :, :2])
    rb = torch.min(x1[:, None, 2:4], x2[None, :, 2:4])
    wh = (rb - lt).clamp(min=0)
    ov = wh[..., 0] * wh[..., 1]
    union = a1[:, None] + a2[None, :] - ov if mode == "iou" else a1[:, None].expand_as(ov)
    want = ov / torch.max(union, union.new_tensor(1e-6))
    assert got.shape == want.shape and torch.equal(got, want)
