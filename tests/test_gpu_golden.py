"""GPU parity against the committed golden vectors (tests/golden/*.npz, produced from the reference's
own sources -- see tests/golden/make_golden.py), through the C ABI."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
GEOM_KEYS = ("B", "C", "H", "W", "kh", "kw", "ph", "pw", "sh", "sw", "dh", "dw", "dg")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("v", [0, 1])
def test_iou_golden(cuda, v):
    from rs_detection_amd.ops.box_iou_rotated import _iou
    d = np.load(os.path.join(G, "iou_v%d.npz" % v))
    for a, b, want in ((d["boxes1"], d["boxes2"], d["ious"]), (d["gts"], d["anchors"], d["ious_anchor"])):
        got = _iou(_t(a, cuda), _t(b, cuda), v).cpu().numpy()
        bad = np.argwhere(~(np.abs(got - want) <= 1e-4))
        assert len(bad) == 0, "%d of %s pairs off by > 1e-4, first %s got %s want %s" % (
            len(bad), got.shape, bad[:4].tolist(), got[tuple(bad[:4].T)], want[tuple(bad[:4].T)])
        assert ((got == 0) == (want == 0)).all()
        assert (got.view(np.int32) == want.view(np.int32)).mean() > 0.999
        for _ in range(5):  # same launch again: bitwise repeatable
            again = _iou(_t(a, cuda), _t(b, cuda), v).cpu().numpy()
            assert (again.view(np.int32) == got.view(np.int32)).all()


@pytest.mark.parametrize("bl", [5, 6])
def test_nms_golden_keep_exact(cuda, bl):
    from rs_detection_amd.ops import nms_rotated_keep_mask
    d = np.load(os.path.join(G, "nms%d.npz" % bl))
    for thr in (0.1, 0.3, 0.8):
        got = nms_rotated_keep_mask(_t(d["dets"], cuda), _t(d["order"], cuda), thr, bl).cpu().numpy()
        assert (got == d["keep_%g" % thr]).all()
    got = nms_rotated_keep_mask(_t(d["known_dets"], cuda), _t(d["known_order"], cuda), 0.3, bl).cpu().numpy()
    assert (got == d["known_keep"]).all()


def test_arf_golden(cuda):
    from rs_detection_amd.ops import arf_forward, arf_backward
    d = np.load(os.path.join(G, "arf.npz"))
    for t in "abc":
        assert (arf_forward(_t(d[t + "_w"], cuda), _t(d[t + "_idx"], cuda)).cpu().numpy() == d[t + "_fwd"]).all()
        assert (arf_backward(_t(d[t + "_idx"], cuda), _t(d[t + "_go"], cuda)).cpu().numpy() == d[t + "_bwd"]).all()


def test_dcn_golden(cuda):
    from rs_detection_amd import ops
    d = np.load(os.path.join(G, "dcn.npz"))
    for t in "abc":
        g = dict(zip(GEOM_KEYS, d[t + "_geom"].tolist()))
        k, p, s, dl = (g["kh"], g["kw"]), (g["ph"], g["pw"]), (g["sh"], g["sw"]), (g["dh"], g["dw"])
        im, off, gcol = _t(d[t + "_im"], cuda), _t(d[t + "_off"], cuda), _t(d[t + "_gcol"], cuda)
        col = ops.deformable_im2col(im, off, k, p, s, dl, g["dg"]).cpu().numpy()
        assert np.abs(col - d[t + "_col"].reshape(col.shape)).max() <= 1e-4
        gcol2 = gcol.reshape(col.shape)
        gim = ops.deformable_col2im(gcol2, off, im.shape, k, p, s, dl, g["dg"]).cpu().numpy()
        assert np.abs(gim - d[t + "_gim"]).max() <= 1e-4 * max(1, np.abs(d[t + "_gim"]).max())
        goff = ops.deformable_col2im_coord(gcol2, im, off, k, p, s, dl, g["dg"]).cpu().numpy()
        assert np.abs(goff - d[t + "_goff"]).max() <= 1e-4 * max(1, np.abs(d[t + "_goff"]).max())


def test_rroi_golden(cuda):
    from rs_detection_amd.ops import roi_align_rotated_v1
    d = np.load(os.path.join(G, "rroi.npz"))
    for t in "abc":
        sc, sr = d[t + "_cfg"]
        feat = _t(d[t + "_feat"], cuda).requires_grad_(True)
        out = roi_align_rotated_v1(feat, _t(d[t + "_rois"], cuda), (7, 7), float(sc), int(sr))
        assert np.abs(out.detach().cpu().numpy() - d[t + "_out"]).max() <= 1e-4
        out.backward(_t(d[t + "_go"], cuda))
        assert np.abs(feat.grad.cpu().numpy() - d[t + "_gfeat"]).max() <= 1e-4 * max(1, np.abs(d[t + "_gfeat"]).max())


def test_rroi_v0_golden(cuda):
    from rs_detection_amd.ops.roi_align_rotated import roi_align
    d = np.load(os.path.join(G, "rroi_v0.npz"))
    for t in "abc":
        sc, sr = d[t + "_cfg"]
        feat = _t(d[t + "_feat"], cuda).requires_grad_(True)
        out = roi_align(feat, _t(d[t + "_rois"], cuda), (7, 7), float(sc), int(sr))
        assert np.abs(out.detach().cpu().numpy() - d[t + "_out"]).max() <= 1e-4
        out.backward(_t(d[t + "_go"], cuda))
        assert np.abs(feat.grad.cpu().numpy() - d[t + "_gfeat"]).max() <= 1e-4 * max(1, np.abs(d[t + "_gfeat"]).max())


def test_feature_refine_golden(cuda):
    from rs_detection_amd.ops.fr import feature_refine
    d = np.load(os.path.join(G, "fr.npz"))
    for t in "abc":
        sc, pt = d[t + "_cfg"]
        feat = _t(d[t + "_feat"], cuda).requires_grad_(True)
        out = feature_refine(feat, _t(d[t + "_boxes"], cuda), float(sc), int(pt))
        assert np.abs(out.detach().cpu().numpy() - d[t + "_out"]).max() <= 1e-4
        out.backward(_t(d[t + "_go"], cuda))
        assert np.abs(feat.grad.cpu().numpy() - d[t + "_gin"]).max() <= 1e-4 * max(1, np.abs(d[t + "_gin"]).max())


def test_convex_sort_golden(cuda):
    """Hull indices are integer work: bit-exact against the reference's CPU loop (fixture), stale slots included."""
    from rs_detection_amd.ops.convex_sort import convex_sort
    d = np.load(os.path.join(G, "convex.npz"))
    for t in "abcde":
        got = convex_sort(_t(d[t + "_pts"], cuda), torch.from_numpy(d[t + "_masks"]).to(cuda),
                          bool(d[t + "_circular"]))
        assert got.dtype == torch.int32 and (got.cpu().numpy() == d[t + "_index"]).all(), t


def test_poly_nms_golden(cuda):
    """fp32 in-model polygon NMS: IoU values bit-identical to the reference arithmetic (fixture), keep lists equal."""
    from rs_detection_amd.ops import poly_nms, poly_iou_f32
    d = np.load(os.path.join(G, "poly_nms.npz"))
    for t in "abc":
        dets = d[t + "_dets"]
        order = np.argsort(-dets[:, 8], kind="stable")
        p = _t(np.ascontiguousarray(dets[order][:96, :8]), cuda)
        got = poly_iou_f32(p, p).cpu().numpy()
        assert (got == d[t + "_iou_sorted"]).all(), (t, np.abs(got - d[t + "_iou_sorted"]).max())
        for thr in (0.1, 0.5):
            keep = poly_nms(_t(dets, cuda), thr).cpu().numpy()
            want = d["%s_keep_%g" % (t, thr)]
            assert len(keep) == len(want) and (keep == want).all(), (t, thr)


def test_assign_and_coder_golden(cuda):
    from rs_detection_amd import ops
    d = np.load(os.path.join(G, "assign.npz"))
    K = d["gts"].shape[0]
    ro = torch.tensor([0, K], dtype=torch.int32, device=cuda)
    ov = ops.box_iou_rotated_grouped(_t(d["gts"], cuda), ro, K, _t(d["anchors"], cuda))
    gi, mo, lb = ops.assign_wrt_overlaps(ov, ro, K, 0.5, 0.4, 0.0, True, True, _t(d["gt_labels"], cuda), 0)
    assert (gi[0].cpu().numpy() == d["gt_inds"]).all()          # bit-exact anchor indices
    assert (lb[0].cpu().numpy() == d["labels"]).all()
    assert np.abs(mo[0].cpu().numpy() - d["max_overlaps"]).max() <= 1e-4
    c = np.load(os.path.join(G, "coder.npz"))
    enc = ops.bbox2delta_rotated(_t(c["proposals"], cuda), _t(c["gt"], cuda)).cpu().numpy()
    assert np.abs((enc - c["encoded"]) / np.maximum(np.abs(c["encoded"]), 1)).max() <= 1e-4
    dec = ops.delta2bbox_rotated(_t(c["proposals"], cuda), _t(c["deltas"], cuda)).cpu().numpy()
    assert np.abs((dec - c["decoded"]) / np.maximum(np.abs(c["decoded"]), 1)).max() <= 1e-4


def test_rie_golden_and_autograd(cuda):
    """HIP rotation-invariant encoding == the reference CPU source's fixtures, bit for bit."""
    from rs_detection_amd.ops import rie_forward, rie_backward, RotationInvariantEncoding
    d = np.load(os.path.join(G, "rie.npz"))
    for tag in "abc":
        nori = int(d[tag + "_nori"])
        direction, aligned = rie_forward(_t(d[tag + "_f"], cuda), nori)
        assert (direction.cpu().numpy() == d[tag + "_dir"]).all()
        assert (aligned.cpu().numpy().view(np.int32) == d[tag + "_aligned"].view(np.int32)).all()
        gi = rie_backward(direction, _t(d[tag + "_go"], cuda), nori)
        assert (gi.cpu().numpy().view(np.int32) == d[tag + "_gi"].view(np.int32)).all()
    x = _t(d["c_f"], cuda).requires_grad_(True)
    out, dirs = RotationInvariantEncoding(8, return_direction=True)(x)
    assert dirs.dtype == torch.uint8 and not dirs.requires_grad
    out.backward(_t(d["c_go"], cuda))
    assert (x.grad.cpu().numpy().view(np.int32) == d["c_gi"].view(np.int32)).all()
