"""CPU: Oriented-RCNN host logic -- anchors, coders, box conversions, samplers, config."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import dota_boxes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_anchor_generator_docstring_known_answers():
    """The two examples in the reference docstring (models/boxes/anchor_generator.py:122-139)."""
    from rs_detection_amd.models.boxes import AnchorGenerator
    g = AnchorGenerator([16], [1.], [1.], [9])
    a = g.grid_anchors([(2, 2)])
    want = torch.tensor([[-4.5, -4.5, 4.5, 4.5], [11.5, -4.5, 20.5, 4.5], [-4.5, 11.5, 4.5, 20.5], [11.5, 11.5, 20.5, 20.5]])
    assert torch.equal(a[0], want)
    g = AnchorGenerator([16, 32], [1.], [1.], [9, 18])
    a = g.grid_anchors([(2, 2), (1, 1)])
    assert torch.equal(a[0], want) and torch.equal(a[1], torch.tensor([[-9., -9., 9., 9.]]))


def test_orcnn_anchor_count_and_valid_flags():
    from rs_detection_amd.models.boxes import AnchorGenerator
    g = AnchorGenerator(strides=[4, 8, 16, 32, 64], ratios=[0.125, 0.25, 0.5, 1.0, 2.0, 4.0, 8.0], scales=[8])
    sizes = [(256, 256), (128, 128), (64, 64), (32, 32), (16, 16)]
    assert sum(a.shape[0] for a in g.grid_anchors(sizes)) == 611072
    assert g.num_base_anchors == [7] * 5
    b = g.base_anchors[0]
    assert torch.allclose((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]), torch.full((7,), 32.0 * 32.0), rtol=1e-5)
    fl = g.valid_flags([(4, 4)] + [(1, 1)] * 4, (12, 12))
    assert fl[0].view(4, 4, 7)[..., 0].sum() == 9


def test_bbox_transforms_roundtrips():
    from rs_detection_amd.ops import bbox_transforms as bt
    rng = np.random.default_rng(0)
    obb = torch.from_numpy(dota_boxes(rng, 200))
    obb[:, 4] = torch.from_numpy(rng.uniform(-math.pi / 2, math.pi / 2 - 1e-3, 200).astype(np.float32))
    poly = bt.obb2poly(obb)
    back = bt.rectpoly2obb(poly)
    assert torch.allclose(back[:, :4], obb[:, :4], atol=1e-2)
    assert torch.allclose(torch.sin(2 * (back[:, 4] - obb[:, 4])), torch.zeros(200), atol=1e-3)
    hbb = bt.obb2hbb(obb)
    assert torch.allclose(hbb, bt.poly2hbb(poly), atol=1e-3)
    assert bt.get_bbox_type(hbb) == 'hbb' and bt.get_bbox_type(obb) == 'obb' and bt.get_bbox_type(poly) == 'poly'
    assert bt.bbox2type(obb, 'hbb').shape == (200, 4) and bt.bbox2type(hbb, 'poly').shape == (200, 8)
    assert torch.allclose(bt.hbb2obb(torch.tensor([[0., 0., 2., 4.]])), torch.tensor([[1.0, 2.0, 4.0, 2.0, -math.pi / 2]]))
    r = bt.regular_theta(torch.tensor([-2.0, 0.3, 4.0]))
    assert (r >= -math.pi / 2).all() and (r < math.pi / 2).all()
    with pytest.raises(ValueError):
        bt.get_bbox_dim('xyz')


def test_oriented_delta_coder_roundtrip():
    from rs_detection_amd.models.boxes import OrientedDeltaXYWHTCoder
    from rs_detection_amd.ops.bbox_transforms import regular_obb
    rng = np.random.default_rng(1)
    prop = regular_obb(torch.from_numpy(dota_boxes(rng, 300, 400)))
    gt = prop.clone()
    gt[:, :2] += torch.from_numpy(rng.normal(0, 5, (300, 2)).astype(np.float32))
    gt[:, 2:4] *= torch.from_numpy(np.exp(rng.normal(0, 0.2, (300, 2))).astype(np.float32))
    gt[:, 4] += torch.from_numpy(rng.normal(0, 0.2, 300).astype(np.float32))
    gt = regular_obb(gt)
    coder = OrientedDeltaXYWHTCoder(target_stds=[0.1, 0.1, 0.2, 0.2, 0.1])
    back = coder.decode(prop, coder.encode(prop, gt))
    assert torch.allclose(back[:, :2], gt[:, :2], atol=1e-2)
    # same rectangle (w/h may swap with a 90-degree turn): compare corner sets through the polygon area centre
    assert torch.allclose(back[:, 2] * back[:, 3], gt[:, 2] * gt[:, 3], rtol=1e-3)


def test_midpoint_offset_coder_roundtrip():
    from rs_detection_amd.models.boxes import MidpointOffsetCoder
    from rs_detection_amd.ops.bbox_transforms import obb2hbb, obb2poly, regular_obb
    rng = np.random.default_rng(2)
    gt = regular_obb(torch.from_numpy(dota_boxes(rng, 200, 400, wmin=40)))
    anchors = obb2hbb(gt) + torch.from_numpy(rng.normal(0, 1, (200, 4)).astype(np.float32))
    coder = MidpointOffsetCoder(target_stds=[1, 1, 1, 1, 0.5, 0.5])
    dec = coder.decode(anchors, coder.encode(anchors, gt))
    assert dec.shape == (200, 5)
    assert torch.allclose(dec[:, :2], gt[:, :2], atol=5e-2)
    # the midpoint-offset representation is exact for rectangles up to the 0.1-px tie rule of :349-355
    rel = ((dec[:, 2] * dec[:, 3]) / (gt[:, 2] * gt[:, 3]) - 1).abs()
    assert (rel < 2e-2).float().mean() > 0.97, rel.max()
    assert ((obb2hbb(dec) - obb2hbb(gt)).abs().max(1)[0] < 0.5).float().mean() > 0.97


def test_random_sampler_counts_and_gt_injection():
    from rs_detection_amd.models.boxes import RandomSamplerRotated, RandomSampler, AssignResult
    torch.manual_seed(0)
    n = 1000
    gt_inds = torch.zeros(n, dtype=torch.int32)
    gt_inds[:300] = torch.randint(1, 6, (300,), dtype=torch.int32)
    gt_inds[900:] = -1
    ar = AssignResult(5, gt_inds, torch.rand(n), labels=torch.randint(0, 10, (n,)))
    boxes, gts = torch.rand(n, 6), torch.rand(5, 5)
    res = RandomSamplerRotated(512, 0.25, add_gt_as_proposals=True).sample(ar, boxes, gts, torch.arange(5))
    assert res.pos_inds.numel() == 128 and res.neg_inds.numel() == 384
    assert res.bboxes.shape == (512, 5) and res.pos_gt_bboxes.shape == (128, 5)
    assert (res.pos_assigned_gt_inds >= 0).all() and res.pos_is_gt.sum() <= 5
    ar2 = AssignResult(5, gt_inds.clone(), torch.rand(n))
    res2 = RandomSampler(256, 0.5, add_gt_as_proposals=False).sample(ar2, torch.rand(n, 4), torch.rand(5, 4))
    assert res2.pos_inds.numel() == 128 and res2.neg_inds.numel() == 128


def test_orcnn_config_matches_reference_and_builds():
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.config import Config
    from rs_detection_amd.utils.registry import MODELS, build_from_cfg
    mine = Config(os.path.join(ROOT, "configs", "orcnn", "orcnn_van3_7_anchor.py"))
    ref = "/root/reference/configs/orcnn_van3_7_anchor_swa_1.py"
    if os.path.exists(ref):
        theirs = Config(ref)
        for k in ("model", "optimizer", "scheduler", "optimizer_swa", "scheduler_swa", "max_epoch"):
            assert mine.dump()[k] == theirs.dump()[k], k
    cfg = mine.dump()["model"]
    cfg["backbone"] = dict(type="van_b0", img_size=256, num_stages=4, out_indices=(0, 1, 2, 3))
    cfg["neck"]["in_channels"] = [32, 64, 160, 256]
    m = build_from_cfg(cfg, MODELS)
    sd = m.state_dict()
    for k in ("backbone.patch_embed1.proj.weight", "backbone.block1.0.attn.spatial_gating_unit.conv_spatial.weight",
              "backbone.block1.0.layer_scale_1", "backbone.norm4.weight", "rpn.rpn_reg.weight",
              "bbox_head.shared_fcs.0.weight", "bbox_head.fc_cls.weight", "bbox_head.fc_reg.bias"):
        assert k in sd, k
    assert tuple(sd["rpn.rpn_cls.weight"].shape) == (7, 256, 1, 1) and tuple(sd["rpn.rpn_reg.weight"].shape) == (42, 256, 1, 1)
    assert tuple(sd["bbox_head.fc_cls.weight"].shape) == (11, 1024) and tuple(sd["bbox_head.shared_fcs.0.weight"].shape) == (1024, 12544)
    feats = m.neck(m.backbone(torch.randn(1, 3, 128, 128)))
    assert [tuple(f.shape[-2:]) for f in feats] == [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2)]


def test_roi_level_mapping():
    from rs_detection_amd.models.roi_extractors.oriented_single_level import OrientedSingleRoIExtractor
    ex = OrientedSingleRoIExtractor(dict(type='ROIAlignRotated_v1', output_size=7, sampling_ratio=2), 256,
                                    [4, 8, 16, 32], extend_factor=(1.4, 1.2))
    rois = torch.tensor([[0, 0, 0, 10, 10, 0], [0, 0, 0, 60, 60, 0], [0, 0, 0, 120, 120, 0], [0, 0, 0, 900, 900, 0.]])
    assert ex.map_roi_levels(rois, 4).tolist() == [0, 0, 1, 3]
    r2 = ex.roi_rescale(rois, (1.4, 1.2))
    assert torch.allclose(r2[:, 3], rois[:, 3] * 1.2) and torch.allclose(r2[:, 4], rois[:, 4] * 1.4)  # q18
