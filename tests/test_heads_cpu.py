"""CPU: known answers for the Oriented R-CNN coders and box-type helpers -- the torch classes of the product against
NumPy transcriptions of the reference (oracle/heads.py, line by line from models/boxes/coder.py:327-513 and
ops/bbox_transforms.py:501-640).  Not round trips: encode and decode are each compared with an independent
restatement on seeded inputs; the round trip is asserted on top."""
import numpy as np
import pytest
import torch

from oracle import heads as H
from conftest import dota_boxes


def _obbs(rng, n, le90=True):
    b = dota_boxes(rng, n, 512)
    if le90:
        b[:, 4] = rng.uniform(-np.pi / 2, np.pi / 2, n)
    return b.astype(np.float32)


def test_box_type_helpers_match_the_transcriptions():
    from rs_detection_amd.ops import bbox_transforms as T
    rng = np.random.default_rng(0)
    obb = _obbs(rng, 200)
    t = torch.from_numpy(obb)
    np.testing.assert_allclose(T.obb2poly(t).numpy(), H.np_obb2poly(obb), rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(T.obb2hbb(t).numpy(), H.np_obb2hbb(obb), rtol=1e-6, atol=1e-4)
    th = rng.uniform(-10, 10, 500).astype(np.float32)
    np.testing.assert_allclose(T.regular_theta(torch.from_numpy(th)).numpy(), H.np_regular_theta(th), atol=1e-6)
    assert (H.np_regular_theta(th) >= -np.pi / 2 - 1e-6).all() and (H.np_regular_theta(th) < np.pi / 2 + 1e-6).all()
    tall = obb.copy()
    tall[:, [2, 3]] = tall[:, [3, 2]]                               # h > w: regular_obb must swap and turn
    np.testing.assert_allclose(T.regular_obb(torch.from_numpy(tall)).numpy(), H.np_regular_obb(tall), atol=1e-5)
    assert (H.np_regular_obb(tall)[:, 2] >= H.np_regular_obb(tall)[:, 3]).all()
    # rectpoly2obb(obb2poly(b)) gives b back (up to the regular form): known answer for the pair
    back = H.np_rectpoly2obb(H.np_obb2poly(obb))
    np.testing.assert_allclose(T.rectpoly2obb(T.obb2poly(t)).numpy(), back, atol=2e-3)
    np.testing.assert_allclose(back[:, :4], H.np_regular_obb(obb)[:, :4], atol=2e-3)


def test_midpoint_offset_coder_known_answers():
    from rs_detection_amd.models.boxes.coder import MidpointOffsetCoder
    rng = np.random.default_rng(1)
    means, stds = (0.01, -0.02, 0.0, 0.03, 0.0, 0.01), (1.0, 1.0, 1.0, 1.0, 0.5, 0.5)
    coder = MidpointOffsetCoder(means, stds)
    gt = _obbs(rng, 300)
    c = gt[:, :2] + rng.normal(0, 6, (300, 2)).astype(np.float32)
    wh = np.abs(gt[:, 2:4]) * np.exp(rng.normal(0, 0.3, (300, 2))).astype(np.float32) + 4
    prop = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)           # hbb proposals
    enc = coder.encode(torch.from_numpy(prop), torch.from_numpy(gt)).numpy()
    want = H.np_midpoint_offset_encode(prop, gt, means, stds)
    np.testing.assert_allclose(enc, want, rtol=1e-5, atol=1e-5)
    # hand-checkable case: axis-aligned gt == proposal -> centre / size deltas 0, midpoint offsets da = +-0.5, db = +-0.5
    g0 = np.array([[50., 40., 20., 10., 0.]], np.float32)
    p0 = np.array([[40., 35., 60., 45.]], np.float32)
    e0 = H.np_midpoint_offset_encode(p0, g0, (0,) * 6, (1,) * 6)[0]
    np.testing.assert_allclose(e0[:4], 0, atol=1e-6)
    assert abs(abs(e0[4]) - 0.5) < 1e-6 and abs(abs(e0[5]) - 0.5) < 1e-6
    # decode
    pred = rng.normal(0, 0.3, (300, 6)).astype(np.float32)
    dec = coder.decode(torch.from_numpy(prop), torch.from_numpy(pred)).numpy()
    np.testing.assert_allclose(dec, H.np_midpoint_offset_decode(prop, pred, means, stds), rtol=1e-4, atol=2e-3)
    # ... and the round trip on top (only exact for gts whose midpoint offsets are inside the decode clamp)
    rt = H.np_midpoint_offset_decode(prop, want, means, stds)
    ok = (np.abs(want[:, 4] * 0.5 + means[4]) < 0.49) & (np.abs(want[:, 5] * 0.5 + means[5]) < 0.49)
    assert ok.sum() > 50
    np.testing.assert_allclose(H.np_obb2hbb(rt[ok]), H.np_obb2hbb(gt[ok]), atol=0.05)


def test_oriented_delta_coder_known_answers():
    from rs_detection_amd.models.boxes.coder import OrientedDeltaXYWHTCoder
    rng = np.random.default_rng(2)
    means, stds = (0., 0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2, 0.1)
    coder = OrientedDeltaXYWHTCoder(means, stds)
    gt, prop = _obbs(rng, 300), _obbs(rng, 300)
    prop[:, :2] = gt[:, :2] + rng.normal(0, 5, (300, 2)).astype(np.float32)
    enc = coder.encode(torch.from_numpy(prop), torch.from_numpy(gt)).numpy()
    want = H.np_oriented_delta_encode(prop, gt, means, stds)
    np.testing.assert_allclose(enc, want, rtol=1e-4, atol=1e-4)
    # hand-checkable: same box -> all zero; gt turned by 90 degrees with (w, h) swapped is the SAME rectangle -> zero
    b = np.array([[10., 20., 30., 8., 0.2]], np.float32)
    np.testing.assert_allclose(H.np_oriented_delta_encode(b, b, means, stds), 0, atol=1e-5)
    b90 = np.array([[10., 20., 8., 30., 0.2 - np.pi / 2]], np.float32)
    np.testing.assert_allclose(H.np_oriented_delta_encode(b, b90, means, stds), 0, atol=1e-4)
    pred = rng.normal(0, 1.0, (300, 5)).astype(np.float32)
    dec = coder.decode(torch.from_numpy(prop), torch.from_numpy(pred)).numpy()
    np.testing.assert_allclose(dec, H.np_oriented_delta_decode(prop, pred, means, stds), rtol=1e-4, atol=2e-3)
    rt = H.np_oriented_delta_decode(prop, want, means, stds)
    np.testing.assert_allclose(H.np_obb2poly(rt).reshape(-1, 4, 2).mean(1), gt[:, :2], atol=1e-2)
    np.testing.assert_allclose(np.sort(rt[:, 2:4], 1), np.sort(gt[:, 2:4], 1), rtol=1e-3, atol=1e-2)


def test_multiclass_nms_glue_known_answer():
    """ops/nms_rotated.py:540-596 restated: two overlapping boxes of one class -> the better survives; the same two
    boxes in DIFFERENT classes both survive; below score_thr nothing; order = score descending."""
    boxes = np.array([[50, 50, 40, 20, 0.1], [52, 50, 40, 20, 0.1], [200, 200, 30, 30, 0.0]], np.float32)
    scores = np.zeros((3, 4), np.float32)
    scores[0, 1], scores[1, 1], scores[2, 2] = 0.9, 0.8, 0.7
    d, l = H.np_multiclass_nms_rotated(boxes, scores, 0.05, 0.1, 100)
    assert d.shape == (2, 6) and list(l) == [0, 1] and np.allclose(d[:, 5], [0.9, 0.7])
    scores[1] = 0
    scores[1, 3] = 0.8
    d, l = H.np_multiclass_nms_rotated(boxes, scores, 0.05, 0.1, 100)
    assert list(l) == [0, 2, 1] and np.allclose(d[:, 5], [0.9, 0.8, 0.7])
    d, l = H.np_multiclass_nms_rotated(boxes, scores, 0.95, 0.1, 100)
    assert d.shape == (0, 6) and l.shape == (0,)


def test_anchor_generator_and_hbb_overlaps_match_the_transcriptions():
    """AnchorGenerator grid / valid flags / inside flags and BboxOverlaps2D (pure torch in the product) against the
    NumPy restatements of models/boxes/anchor_generator.py:94-470, anchor_target.py:184-195, iou_calculator.py:164-270."""
    from rs_detection_amd.models.boxes.anchor_generator import AnchorGenerator
    from rs_detection_amd.models.boxes.anchor_target import anchor_inside_flags
    from rs_detection_amd.models.boxes.iou_calculator import bbox_overlaps
    from rs_detection_amd.ops.bbox_transforms import hbb2obb
    strides, ratios, scales = [4, 8, 16, 32, 64], [0.125, 0.25, 0.5, 1.0, 2.0, 4.0, 8.0], [8]
    sizes = [(24, 32), (12, 16), (6, 8), (3, 4), (2, 2)]
    ag = AnchorGenerator(strides=strides, ratios=ratios, scales=scales)
    got = ag.grid_anchors(sizes)
    want = H.np_anchor_generator_grid(strides, ratios, scales, sizes)
    for g, w in zip(got, want):
        np.testing.assert_allclose(g.numpy(), w, rtol=1e-6, atol=1e-4)
    # the doc example of the reference (anchor_generator.py:122-128): stride 16, ratio 1, scale 1, base size 9 is not
    # reachable through np_anchor_generator_grid (base = stride); check the stride-16 / scale-1 analogue by hand
    a = H.np_anchor_generator_grid([16], [1.0], [1.0], [(2, 2)])[0]
    np.testing.assert_allclose(a, [[-8, -8, 8, 8], [8, -8, 24, 8], [-8, 8, 8, 24], [8, 8, 24, 24]])
    pad = (90, 100)
    vf = ag.valid_flags(sizes, pad)
    wf = H.np_anchor_valid_flags(strides, sizes, pad, 7)
    for g, w in zip(vf, wf):
        assert (g.numpy() == w).all()
    assert 0 < wf[0].sum() < wf[0].size
    flat, valid = torch.cat(got), torch.cat(vf)
    ins = anchor_inside_flags(flat, valid, (96, 128), allowed_border=0).numpy()
    assert (ins == H.np_anchor_inside_flags(flat.numpy(), valid.numpy(), (96, 128), 0)).all() and 0 < ins.sum() < ins.size
    rng = np.random.default_rng(3)
    c, wh = rng.uniform(0, 100, (40, 2)), rng.uniform(2, 40, (40, 2))
    b1 = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    b2 = flat.numpy()[::37]
    np.testing.assert_allclose(bbox_overlaps(torch.from_numpy(b1), torch.from_numpy(b2)).numpy(),
                               H.np_bbox_overlaps_hbb(b1, b2), rtol=1e-6, atol=1e-7)
    # known answers of the reference's docstring (iou_calculator.py:188-199)
    d1 = np.array([[0, 0, 10, 10], [10, 10, 20, 20], [32, 32, 38, 42]], np.float32)
    d2 = np.array([[0, 0, 10, 20], [0, 10, 10, 19], [10, 10, 20, 20]], np.float32)
    np.testing.assert_allclose(H.np_bbox_overlaps_hbb(d1, d2), [[0.5, 0, 0], [0, 0, 1], [0, 0, 0]], atol=1e-6)
    np.testing.assert_allclose(hbb2obb(torch.from_numpy(b1)).numpy(), H.np_hbb2obb(b1), atol=1e-5)


def test_random_sample_restatement_on_a_fixed_choice():
    """sampler.py:57-111 with a fixed choice: counts, disjointness, the neg_pos_ub bound, sortedness (`unique`)."""
    rng = np.random.default_rng(4)
    gi = rng.choice([-1, 0, 0, 0, 1, 2, 3], 5000).astype(np.int32)
    first = lambda g, n: g[:n]
    pos, neg = H.np_random_sample(gi, 256, 0.5, -1, first)
    assert len(pos) == 128 and len(neg) == 128 and (gi[pos] > 0).all() and (gi[neg] == 0).all()
    assert (np.diff(pos) > 0).all() and (np.diff(neg) > 0).all()
    few = np.zeros(1000, np.int32)
    few[[5, 50]] = 1
    pos, neg = H.np_random_sample(few, 256, 0.5, -1, first)
    assert list(pos) == [5, 50] and len(neg) == 254
    pos, neg = H.np_random_sample(few, 256, 0.5, 3, first)
    assert len(neg) == 6
