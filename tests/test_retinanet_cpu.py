"""BASELINE config[0]: RetinaNet-hbb R50-FPN -- the reference's CPU-runnable plumbing case (no rotated op, no GPU).
The reference's own mode 'H' branch is unfinished (retina_head.py:35 "#TODO: check 'H' mode"), so there is no parity
target: this proves config -> registry -> RetinaNet -> Resnet50 -> FPN -> RetinaHead builds and trains on CPU, and
pins the helper arithmetic (box_ops.py:5-129, anchor_generator.py:495-640, retinanet.py:33-45) against NumPy twins."""
import os

import numpy as np
import torch

import rs_detection_amd.models  # noqa: F401
import rs_detection_amd.optims  # noqa: F401  (registers SGD / AdamW / schedulers)
from rs_detection_amd.config import init_cfg, get_cfg
from rs_detection_amd.utils.registry import build_from_cfg, MODELS, BOXES, OPTIMS
from rs_detection_amd.models.boxes import box_ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _targets(rng, n, size, K=6):
    out = []
    for _ in range(n):
        c = rng.uniform(0.15 * size, 0.85 * size, (K, 2))
        wh = rng.uniform(0.08 * size, 0.4 * size, (K, 2))
        hb = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
        out.append(dict(hboxes=torch.from_numpy(hb), labels=torch.from_numpy(rng.integers(1, 16, K).astype(np.int32)),
                        img_size=(size, size)))
    return out


def test_retinanet_hbb_config_builds_and_trains_on_cpu():
    init_cfg(os.path.join(ROOT, "configs/retinanet/retinanet_hbb_r50_fpn.py"))
    cfg = get_cfg()
    torch.manual_seed(0)
    model = build_from_cfg(cfg.model, MODELS)
    assert type(model).__name__ == "RetinaNet" and type(model.rpn_net).__name__ == "RetinaHead"
    assert model.rpn_net.retina_reg.out_channels == 9 * 4 and model.rpn_net.retina_cls.out_channels == 9 * 15
    opt = build_from_cfg(cfg.optimizer, OPTIMS, params=[p for p in model.parameters() if p.requires_grad])
    model.train()
    rng = np.random.default_rng(0)
    images = torch.randn(2, 3, 600, 600)          # BASELINE.json configs[0]: 2 x 600 x 600 synthetic tiles, CPU
    targets = _targets(rng, 2, 600)
    first = None
    for _ in range(2):
        losses = model(images, targets)
        assert set(losses) == {"roi_cls_loss", "roi_loc_loss"}
        total = sum(losses.values())
        assert torch.isfinite(total)
        opt.zero_grad()
        total.backward()
        opt.step()
        first = float(total) if first is None else first
    grads = [p.grad for p in model.rpn_net.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in grads)
    assert model.backbone.conv1.weight.grad is None or not model.backbone.conv1.weight.requires_grad  # frozen_stages=1


def test_retina_anchor_generator_closed_form():
    g = build_from_cfg(dict(type="AnchorGeneratorRotated", strides=[8, 16], ratios=[0.5, 1.0, 2.0],
                            scales=[4., 5.0396842, 6.34960421], mode="H"), BOXES)
    assert g.num_base_anchors == [9, 9]
    a = g.grid_anchors([[3, 5], [2, 2]])
    assert a[0].shape == (3 * 5 * 9, 4) and a[1].shape == (2 * 2 * 9, 4)
    # mode 'H' is scale-major: anchor index = scale * 3 + ratio; centre = stride * (col + 0.5), x fastest
    s, r, col, row, stride = 1, 2, 4, 2, 8
    w = stride * 5.0396842 / np.sqrt(2.0)
    h = stride * 5.0396842 * np.sqrt(2.0)
    got = a[0][(row * 5 + col) * 9 + s * 3 + r].numpy()
    cx, cy = stride * (col + 0.5), stride * (row + 0.5)
    np.testing.assert_allclose(got, [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], rtol=1e-6)
    gr = build_from_cfg(dict(type="AnchorGeneratorRotated", strides=[8], ratios=[0.5, 2.0], scales=[4.],
                             angles=[0., 0.5], mode="R"), BOXES)
    assert gr.grid_anchors([[2, 2]])[0].shape == (2 * 2 * 4, 5)


def test_loc_coders_against_numpy_twin():
    rng = np.random.default_rng(3)
    n = 64
    c = rng.uniform(50, 500, (n, 2))
    wh = rng.uniform(8, 200, (n, 2))
    src = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    c2 = c + rng.normal(0, 10, (n, 2))
    wh2 = wh * np.exp(rng.normal(0, 0.3, (n, 2)))
    dst = np.concatenate([c2 - wh2 / 2, c2 + wh2 / 2], 1).astype(np.float32)
    loc = box_ops.bbox2loc(torch.from_numpy(src), torch.from_numpy(dst)).numpy()
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    dw, dh = dst[:, 2] - dst[:, 0], dst[:, 3] - dst[:, 1]
    want = np.stack([((dst[:, 0] + 0.5 * dw) - (src[:, 0] + 0.5 * sw)) / sw, ((dst[:, 1] + 0.5 * dh) - (src[:, 1] + 0.5 * sh)) / sh,
                     np.log(dw / sw), np.log(dh / sh)], 1)
    np.testing.assert_allclose(loc, want, rtol=2e-5, atol=2e-6)
    back = box_ops.loc2bbox(torch.from_numpy(src), torch.from_numpy(loc)).numpy()
    np.testing.assert_allclose(back, dst, rtol=1e-4, atol=1e-3)       # round trip
    # rotated pair: bbox2loc_r divides by (side + 1) and adds 1e-5 inside the log, loc2bbox_r does not undo that
    a = np.concatenate([c, wh, rng.uniform(-np.pi / 2, 0, (n, 1))], 1).astype(np.float32)
    b = np.concatenate([c2, wh2, rng.uniform(-np.pi / 2, 0, (n, 1))], 1).astype(np.float32)
    lr = box_ops.bbox2loc_r(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    want = np.stack([(b[:, 0] - a[:, 0]) / (a[:, 2] + 1), (b[:, 1] - a[:, 1]) / (a[:, 3] + 1),
                     np.log(b[:, 2] / (a[:, 2] + 1) + 1e-5), np.log(b[:, 3] / (a[:, 3] + 1) + 1e-5), b[:, 4] - a[:, 4]], 1)
    np.testing.assert_allclose(lr, want, rtol=2e-5, atol=2e-6)
    dec = box_ops.loc2bbox_r(torch.from_numpy(a), torch.from_numpy(lr)).numpy()
    np.testing.assert_allclose(dec[:, 4], b[:, 4], atol=1e-6)
    np.testing.assert_allclose(dec[:, 2], np.exp(lr[:, 2]) * a[:, 2], rtol=1e-5)
    # bbox_iou: no +1 convention, zero when disjoint
    iou = box_ops.bbox_iou(torch.tensor([[0., 0., 10., 10.]]), torch.tensor([[5., 0., 15., 10.], [20., 20., 30., 30.]]))
    np.testing.assert_allclose(iou.numpy(), [[50. / 150., 0.]], rtol=1e-6)
    # hull of a rotated box and the x0y0x1y1 <-> xywh pair
    hull = box_ops.rotated_box_to_bbox(torch.tensor([[10., 20., 8., 4., np.pi / 2]])).numpy()
    np.testing.assert_allclose(hull, [[8., 16., 12., 24.]], atol=1e-5)
    x = torch.tensor([[1., 2., 5., 10., 0.3]])
    np.testing.assert_allclose(box_ops.boxes_xywh_to_x0y0x1y1(box_ops.boxes_x0y0x1y1_to_xywh(x)).numpy(), x.numpy(), atol=1e-6)


def test_retinanet_angle_fold_matches_reference_loop():
    from rs_detection_amd.models.networks.retinanet import RetinaNet
    from rs_detection_amd.models.roi_heads.retina_head import RetinaHead
    rng = np.random.default_rng(5)
    gt = np.concatenate([rng.uniform(0, 500, (200, 2)), rng.uniform(5, 100, (200, 2)),
                         rng.uniform(-np.pi / 4, 3 * np.pi / 4, (200, 1))], 1).astype(np.float32)
    out = []
    for x, y, w, h, a in gt:  # retinanet.py:36-45, literally
        if a >= 0:
            a -= np.pi
        if a < -np.pi / 2:
            a += np.pi / 2
            w, h = h, w
        out.append([x, y, w, h, a])
    want = np.array(out, np.float32)
    got = RetinaNet.fold_angles(torch.from_numpy(gt)).numpy()
    np.testing.assert_allclose(got, want, atol=1e-6)
    assert (got[:, 4] >= -np.pi / 2 - 1e-6).all() and (got[:, 4] < 1e-6).all()
    # cvt2_w_greater_than_h(reverse_hw=False): rows with w <= h get (h, w, a + pi/2) and every row a - pi/2
    cv = RetinaHead.cvt2_w_greater_than_h(torch.from_numpy(want), False).numpy()
    keep = want[:, 2] > want[:, 3]
    np.testing.assert_allclose(cv[keep], want[keep] - np.array([0, 0, 0, 0, np.pi / 2], np.float32), atol=1e-6)
    np.testing.assert_allclose(cv[~keep][:, [2, 3]], want[~keep][:, [3, 2]], atol=1e-6)
    np.testing.assert_allclose(cv[~keep][:, 4], want[~keep][:, 4], atol=1e-6)
