"""GPU: RetinaNet in the reference's mode 'R' (projects/retinanet config: horizontal anchors, rotated targets,
5-column regression, per-class rotated NMS through the HIP kernel) -- one train step and one eval pass; and the
mode 'H' config[0] model on the GPU (hbb NMS kernel in eval)."""
import os

import numpy as np
import pytest
import torch

from conftest import dota_boxes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(mode, anchor_mode="H"):
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.utils.registry import build_from_cfg, MODELS
    cfg = dict(
        type="RetinaNet",
        backbone=dict(type='Resnet50', frozen_stages=1, return_stages=["layer1", "layer2", "layer3", "layer4"], pretrained=False),
        neck=dict(type="FPN", in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
                  add_extra_convs="on_input", num_outs=5),
        rpn_net=dict(type="RetinaHead", n_class=15, in_channels=256, stacked_convs=4, mode=mode, score_threshold=0.05,
                     nms_iou_threshold=0.3, max_dets=10000, roi_beta=1 / 9., cls_loss_weight=1., loc_loss_weight=0.2,
                     anchor_generator=dict(type="AnchorGeneratorRotated", strides=[8, 16, 32, 64, 128], ratios=[0.5, 1.0, 2.0],
                                           scales=[4., 5.0396842, 6.34960421], mode=anchor_mode,
                                           **(dict(angles=[0., 0.7]) if anchor_mode == "R" else {}))))
    torch.manual_seed(0)
    return build_from_cfg(cfg, MODELS)


def _targets(rng, dev, n, size, K=12):
    out = []
    for _ in range(n):
        rb = dota_boxes(rng, K, size)
        x, y, w, h, a = rb.T
        cs, sn = np.abs(np.cos(a)), np.abs(np.sin(a))
        hw, hh = (w * cs + h * sn) / 2, (w * sn + h * cs) / 2
        hb = np.stack([x - hw, y - hh, x + hw, y + hh], 1).astype(np.float32)
        out.append(dict(rboxes=torch.from_numpy(rb).to(dev), hboxes=torch.from_numpy(hb).to(dev),
                        labels=torch.from_numpy(rng.integers(1, 16, K).astype(np.int32)).to(dev),
                        img_size=(size, size), ori_img_size=(size, size)))
    return out


@pytest.mark.parametrize("mode,anchor_mode", [("R", "H"), ("R", "R"), ("H", "H")])
def test_retinanet_train_step_and_eval(cuda, mode, anchor_mode):
    model = _model(mode, anchor_mode).to(cuda)
    rng = np.random.default_rng(1)
    images = torch.randn(2, 3, 512, 512, device=cuda)
    targets = _targets(rng, cuda, 2, 512)
    model.train()
    keep_rb = [t["rboxes"].clone() for t in targets]
    losses = model(images, targets)
    assert all(torch.equal(a, t["rboxes"]) for a, t in zip(keep_rb, targets))  # caller's targets untouched
    total = sum(losses.values())
    assert torch.isfinite(total) and float(losses["roi_loc_loss"]) > 0        # some anchors are positive
    total.backward()
    assert all(torch.isfinite(p.grad).all() for p in model.rpn_net.parameters())
    model.eval()
    with torch.no_grad():
        model.rpn_net.score_thresh = 0.009   # sigmoid(prior bias) = 0.01: lets the untrained head emit boxes
        model.rpn_net.nms_pre = 300
        res = model(images, targets)
    assert len(res) == 2
    for polys, scores, labels in res:
        assert polys.shape[1] == 8 and polys.shape[0] == scores.shape[0] == labels.shape[0] > 0
        assert torch.isfinite(polys).all() and labels.dtype == torch.int32
