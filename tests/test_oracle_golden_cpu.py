"""CPU: the oracle restatement vs (a) committed golden vectors produced from the reference's own
sources and (b) oracle/_ref itself when it is present.  No GPU, no /root/reference needed."""
import os

import numpy as np
import pytest

import oracle
from conftest import dota_boxes, degenerate_boxes

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name))


@pytest.mark.parametrize("v", [0, 1])
def test_iou_golden_bit_exact(oracle_c, v):
    d = load("iou_v%d.npz" % v)
    got = oracle_c.box_iou_rotated(d["boxes1"], d["boxes2"], v)
    assert (got.view(np.int32) == d["ious"].view(np.int32)).all()
    got = oracle_c.box_iou_rotated(d["gts"], d["anchors"], v)
    assert (got.view(np.int32) == d["ious_anchor"].view(np.int32)).all()
    np.testing.assert_allclose(oracle_c.box_iou_rotated(d["known_in"], d["known_in"], v), d["known_out"], atol=1e-6)


@pytest.mark.parametrize("bl", [5, 6])
def test_nms_golden_bit_exact(oracle_c, bl):
    d = load("nms%d.npz" % bl)
    for thr in (0.1, 0.3, 0.8):
        assert (oracle_c.nms_rotated(d["dets"], d["order"], thr) == d["keep_%g" % thr]).all()
    assert (oracle_c.nms_rotated(d["known_dets"], d["known_order"], 0.3) == d["known_keep"]).all()
    assert d["known_keep"].tolist() == [False, False, True]  # nms_rotated.py:598-603 smoke input


def test_arf_golden_exact(oracle_c):
    d = load("arf.npz")
    for t in "abc":
        assert (oracle_c.arf_forward(d[t + "_w"], d[t + "_idx"]) == d[t + "_fwd"]).all()
        assert (oracle_c.arf_backward(d[t + "_idx"], d[t + "_go"]) == d[t + "_bwd"]).all()


def test_arf_backward_is_transpose_of_forward(oracle_c):
    """<ARF(w), g> == <w, ARF^T(g)> at the full S2ANet size (beyond the reference CPU kernel's uint16 range)."""
    from rs_detection_amd.ops.orn import arf_indices
    rng = np.random.default_rng(0)
    idx = arf_indices(1, 8, (3, 3)).numpy()
    w = rng.standard_normal((32, 256, 1, 3, 3)).astype(np.float32)
    g = rng.standard_normal((256, 256, 3, 3)).astype(np.float32)
    lhs = float((oracle_c.arf_forward(w, idx).astype(np.float64) * g).sum())
    rhs = float((w.astype(np.float64) * oracle_c.arf_backward(idx, g)).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(1, abs(lhs))


GEOM_KEYS = ("B", "C", "H", "W", "kh", "kw", "ph", "pw", "sh", "sw", "dh", "dw", "dg")


def test_dcn_golden(oracle_c):
    d = load("dcn.npz")
    for t in "abc":
        g = dict(zip(GEOM_KEYS, d[t + "_geom"].tolist()))
        assert np.abs(oracle_c.deform_im2col(d[t + "_im"], d[t + "_off"], g, g["dg"]) - d[t + "_col"]).max() == 0
        gim = oracle_c.deform_col2im(d[t + "_gcol"], d[t + "_off"], d[t + "_im"].shape, g, g["dg"])
        assert np.abs(gim - d[t + "_gim"]).max() <= 1e-6
        goff = oracle_c.deform_col2im_coord(d[t + "_gcol"], d[t + "_im"], d[t + "_off"], g, g["dg"])
        assert np.abs(goff - d[t + "_goff"]).max() <= 1e-5


def test_rroi_golden(oracle_c):
    d = load("rroi.npz")
    for t in "abc":
        sc, sr = d[t + "_cfg"]
        out = oracle_c.rroi_align_v1_forward(d[t + "_feat"], d[t + "_rois"], (7, 7), float(sc), int(sr))
        assert np.abs(out - d[t + "_out"]).max() <= 1e-5
        gf = oracle_c.rroi_align_v1_backward(d[t + "_go"], d[t + "_rois"], d[t + "_feat"].shape, float(sc), int(sr))
        assert np.abs(gf - d[t + "_gfeat"]).max() <= 1e-5


def test_rroi_v0_golden(oracle_c):
    """ROIAlignRotated (ops/roi_align_rotated.py) restatement vs the shimmed reference text (make_golden.make_rroi_v0)."""
    d = load("rroi_v0.npz")
    for t in "abc":
        sc, sr = d[t + "_cfg"]
        out = oracle_c.rroi_align_v1_forward(d[t + "_feat"], d[t + "_rois"], (7, 7), float(sc), int(sr), "v0")
        assert np.abs(out - d[t + "_out"]).max() <= 1e-5
        gf = oracle_c.rroi_align_v1_backward(d[t + "_go"], d[t + "_rois"], d[t + "_feat"].shape, float(sc), int(sr),
                                             "v0")
        assert np.abs(gf - d[t + "_gfeat"]).max() <= 1e-5
        # v0 differs from v1 (guards against the variant flag being dropped on the way)
        assert np.abs(out - oracle_c.rroi_align_v1_forward(d[t + "_feat"], d[t + "_rois"], (7, 7), float(sc),
                                                           int(sr))).max() > 1e-3


def test_feature_refine_golden(oracle_c):
    """FeatureRefine (ops/fr.py) restatement vs the shimmed reference text (make_golden.make_fr): exact."""
    d = load("fr.npz")
    for t in "abc":
        sc, pt = d[t + "_cfg"]
        out = oracle_c.feature_refine_forward(d[t + "_feat"], d[t + "_boxes"], float(sc), int(pt))
        assert np.abs(out - d[t + "_out"]).max() == 0
        gi = oracle_c.feature_refine_backward(d[t + "_go"], d[t + "_boxes"], float(sc), int(pt))
        assert np.abs(gi - d[t + "_gin"]).max() <= 1e-6


def test_convex_sort_golden(oracle_c):
    """convex_sort: the C restatement of the scan == the reference's CPU loop (fixture), bit for bit, including the
    stale stack slots the reference leaves behind; the hull it describes has the area scipy's ConvexHull finds."""
    d = load("convex.npz")
    for t in "abcde":
        pts, m, circ = d[t + "_pts"], d[t + "_masks"], bool(d[t + "_circular"])
        x, y, mf, start, order = oracle.np_convex_sort_prepare(pts, m)
        assert (start == d[t + "_start"]).all() and (order == d[t + "_order"]).all()
        got = oracle_c.convex_sort_scan(x, y, mf, start, order, circ)
        assert (got == d[t + "_index"]).all()
    from scipy.spatial import ConvexHull
    pts, m, idx = d["a_pts"], d["a_masks"], d["a_index"]
    for i in range(60):
        n = int(np.argmax(idx[i, 1:] == idx[i, 0])) + 1   # closing index
        P = pts[i][idx[i, :n]].astype(np.float64)
        area = 0.5 * abs(np.sum(P[:, 0] * np.roll(P[:, 1], -1) - np.roll(P[:, 0], -1) * P[:, 1]))
        if m[i].sum() >= 3:
            assert abs(area - ConvexHull(pts[i][m[i]].astype(np.float64)).volume) <= 1e-3 * area


def test_poly_nms_golden(oracle_c):
    """In-model fp32 polygon NMS (ops/nms_poly.py:17-210): the restated float arithmetic equals the reference's
    devPolyIoU (through the host shim) bit for bit -- including its cancellation noise (IoU < 0 and > 1 in case b) --
    so the keep lists are identical."""
    d = load("poly_nms.npz")
    for t in "abc":
        dets = d[t + "_dets"]
        order = np.argsort(-dets[:, 8], kind="stable")
        p = np.ascontiguousarray(dets[order][:96, :8])
        assert (oracle_c.poly_iou_f32(p, p) == d[t + "_iou_sorted"]).all()
        for thr in (0.1, 0.5):
            want = d["%s_keep_%g" % (t, thr)]
            got = oracle_c.poly_nms(dets, thr)
            assert len(got) == len(want) and (got == want).all()
    assert d["b_iou_sorted"].max() > 1.0 and d["b_iou_sorted"].min() < 0.0   # the noise is really there
    # known answers: unit square vs itself / half-shifted / disjoint, both orientations
    sq = np.array([[0, 0, 1, 0, 1, 1, 0, 1]], np.float32)
    cw = sq.reshape(1, 4, 2)[:, ::-1].reshape(1, 8).copy()
    sh = sq + np.array([0.5, 0] * 4, np.float32)
    far = sq + 5
    iou = oracle_c.poly_iou_f32(np.concatenate([sq, cw]), np.concatenate([sq, sh, far]))
    assert np.abs(iou - np.array([[1, 1 / 3., 0], [1, 1 / 3., 0]], np.float32)).max() < 1e-6


def test_assign_golden(oracle_c):
    d = load("assign.npz")
    gi, mo, lb = oracle_c.assign_wrt_overlaps(d["overlaps"], 0.5, 0.4, 0.0, True, True, d["gt_labels"], 0)
    assert (gi == d["gt_inds"]).all() and (mo == d["max_overlaps"]).all() and (lb == d["labels"]).all()
    # independent NumPy restatement of assigner.py:125-168
    ov = d["overlaps"]
    mx, am = ov.max(0), ov.argmax(0)
    want = np.full(ov.shape[1], -1, np.int32)
    want[(mx >= 0) & (mx < 0.4)] = 0
    want[mx >= 0.5] = am[mx >= 0.5] + 1
    for i in range(ov.shape[0]):
        if ov[i].max() >= 0.0:
            want[ov[i] == ov[i].max()] = i + 1
    assert (gi == want).all()


def test_coder_golden_and_closed_forms():
    d = load("coder.npz")
    np.testing.assert_allclose(oracle.np_bbox2delta_rotated(d["proposals"], d["gt"]), d["encoded"], atol=1e-6)
    np.testing.assert_allclose(oracle.np_delta2bbox_rotated(d["proposals"], d["deltas"]), d["decoded"], atol=1e-4)
    # norm_angle(le135) range and periodicity (box_ops.py:176-182)
    a = np.linspace(-10, 10, 1001).astype(np.float32)
    n = oracle.np_norm_angle(a)
    assert (n >= -np.pi / 4 - 1e-6).all() and (n < 3 * np.pi / 4 + 1e-6).all()
    assert np.abs(np.sin(2 * (n - a))).max() < 1e-4  # differs from the input by a multiple of pi
    # identity delta decodes to the anchor itself (angle normalised)
    p = d["proposals"]
    back = oracle.np_delta2bbox_rotated(p, np.zeros_like(p))
    np.testing.assert_allclose(back[:, :4], p[:, :4], rtol=1e-6)


def test_anchor_grid_closed_form():
    """x = col*s + 0.5(s-1), size 4s, x fastest; 16384/4096/1024/256/64 per level."""
    from conftest import s2anet_anchors
    a = s2anet_anchors()
    assert a.shape == (21824, 5)
    start = 0
    for s in (8, 16, 32, 64, 128):
        f = 1024 // s
        lvl = a[start:start + f * f].reshape(f, f, 5)
        assert (lvl[..., 2] == 4 * s).all() and (lvl[..., 3] == 4 * s).all() and (lvl[..., 4] == 0).all()
        assert (lvl[0, :, 0] == np.arange(f) * s + 0.5 * (s - 1)).all()
        assert (lvl[:, 0, 1] == np.arange(f) * s + 0.5 * (s - 1)).all()
        start += f * f


@pytest.mark.parametrize("v", [0, 1])
def test_oracle_equals_reference_build(oracle_c, oracle_ref, v):
    """Fresh random + degenerate inputs: restatement == reference CPU source, bit for bit."""
    rng = np.random.default_rng(99 + v)
    b1 = np.concatenate([dota_boxes(rng, 150, 250), degenerate_boxes()])
    b2 = np.concatenate([dota_boxes(rng, 900, 250), degenerate_boxes()])
    a, e = oracle_c.box_iou_rotated(b1, b2, v), oracle_ref.box_iou_rotated(b1, b2, v)
    assert (a.view(np.int32) == e.view(np.int32)).all()
    assert (e > 0).mean() > 0.05


def test_oracle_nms_equals_reference_build(oracle_c, oracle_ref):
    rng = np.random.default_rng(5)
    d = dota_boxes(rng, 400, 200)
    s = rng.uniform(0, 1, 400).astype(np.float32)
    order = np.argsort(-s, kind="stable").astype(np.int32)
    for thr in (0.05, 0.5):
        assert (oracle_c.nms_rotated(d, order, thr) == oracle_ref.nms_rotated(d, order, thr)).all()
    d6 = np.concatenate([d, rng.integers(0, 3, (400, 1)).astype(np.float32)], 1)
    assert (oracle_c.nms_rotated(d6, order, 0.2) == oracle_ref.nms_rotated(d6, order, 0.2)).all()


def test_rie_oracle_matches_reference_cpu_source():
    """rotation-invariant encoding, oracle restatement == fixtures from the reference's own CPU source
    (ops/orn.py:290-363), forward (direction + aligned) and backward, bit for bit; first-maximum tie rule."""
    import oracle
    d = np.load(os.path.join(G, "rie.npz"))
    c = oracle.c()
    for tag in "abc":
        nori = int(d[tag + "_nori"])
        direction, aligned = c.rie_forward(d[tag + "_f"], nori)
        assert (direction == d[tag + "_dir"]).all()
        assert (aligned.view(np.int32) == d[tag + "_aligned"].view(np.int32)).all()
        gi = c.rie_backward(d[tag + "_dir"], d[tag + "_go"], nori)
        assert (gi.view(np.int32) == d[tag + "_gi"].view(np.int32)).all()
        assert direction[0, 0] == 0                                   # all-equal group: first orientation
        # the shift brings the maximum to slot 0, and backward inverts forward's permutation
        f = d[tag + "_f"].reshape(direction.shape[0], -1, nori)
        assert (aligned.reshape(f.shape)[:, :, 0] == f.max(-1)).all()
        assert (c.rie_backward(direction, aligned, nori).reshape(f.shape) == f).all()
    assert d["a_dir"][1, 0] == 1                                      # [2, 5, 5, ...]: the first of the two maxima
