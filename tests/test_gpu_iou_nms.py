"""GPU parity: HIP rotated IoU / NMS / assigner (through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import dota_boxes, s2anet_anchors, degenerate_boxes

pytestmark = pytest.mark.gpu
IOU_TOL = 1e-4  # north_star: rotated IoU within 1e-4 fp32


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("version", [0, 1])
def test_iou_known_answer(cuda, version):
    from rs_detection_amd.ops import box_iou_rotated, box_iou_rotated_v1
    b = _t(np.array([[0, 0, 1, 1, 0], [.5, .5, 1, 2, 0]], np.float32), cuda)
    fn = box_iou_rotated if version == 0 else box_iou_rotated_v1
    got = fn(b, b).cpu().numpy()
    np.testing.assert_allclose(got, [[1, .2], [.2, 1]], atol=1e-6)  # box_iou_rotated.py:512-516


@pytest.mark.parametrize("version", [0, 1])
@pytest.mark.parametrize("n1,n2,span", [(1, 1, 50), (7, 300, 200), (100, 3000, 400), (257, 513, 300), (16, 2000, 1024)])
def test_iou_random_vs_oracle(cuda, oracle_c, version, n1, n2, span):
    from rs_detection_amd import ops
    rng = np.random.default_rng(100 * n1 + n2 + version)
    b1, b2 = dota_boxes(rng, n1, span), dota_boxes(rng, n2, span)
    want = oracle_c.box_iou_rotated(b1, b2, version)
    fn = ops.box_iou_rotated if version == 0 else ops.box_iou_rotated_v1
    got = fn(_t(b1, cuda), _t(b2, cuda)).cpu().numpy()
    assert np.abs(got - want).max() <= IOU_TOL
    # disjoint pairs must be EXACT zeros (assigner uses `max_overlaps >= 0 & < neg_thr`)
    assert ((want == 0) == (got == 0)).all()
    # the kernel keeps the reference's op order: expect (near-)bitwise agreement
    same = (got.view(np.int32) == want.view(np.int32)).mean()
    assert same > 0.999, same


@pytest.mark.parametrize("version", [0, 1])
def test_iou_degenerate(cuda, oracle_c, version):
    from rs_detection_amd.ops.box_iou_rotated import _iou
    d = degenerate_boxes()
    want = oracle_c.box_iou_rotated(d, d, version)
    got = _iou(_t(d, cuda), _t(d, cuda), version).cpu().numpy()
    assert np.abs(got - want).max() <= IOU_TOL


@pytest.mark.parametrize("version", [0, 1])
def test_iou_near_collinear_pairs_one_per_launch(cuda, oracle_c, version):
    """Near-parallel boxes sharing edges: the reference's tolerance sort predicate is not transitive on
    their (almost collinear) intersection points.  One pair per launch, so that the clipper's LDS scratch
    holds whatever the previous kernel left there: a fast path that trusted an inconsistent ranking used
    to read a stale slot (NaN / -1.0 on a freshly booted GPU).  Regression: [20,10,20,10,0] x
    [10,10,20,10,1e-4] must give the reference's 6.2644885e-06."""
    from rs_detection_amd.ops.box_iou_rotated import _iou
    rng = np.random.default_rng(11)
    a = [np.array([20., 10., 20., 10., 0.], np.float32)]
    b = [np.array([10., 10., 20., 10., 1e-4], np.float32)]
    for _ in range(40):
        cx, cy, w, h = rng.uniform(5, 50, 4).astype(np.float32)
        th = np.float32(rng.choice([0.0, np.pi / 2, rng.uniform(-3, 3)]))
        eps = np.float32(rng.choice([1e-4, -1e-4, 3e-5, 1e-3]))
        along = np.array([np.cos(th), np.sin(th)], np.float32) * w * np.float32(rng.choice([0.5, 1.0, 0.25]))
        a.append(np.array([cx, cy, w, h, th], np.float32))
        b.append(np.array([cx + along[0], cy + along[1], w, h, th + eps], np.float32))
    a, b = np.stack(a), np.stack(b)
    for i in range(len(a)):
        want = oracle_c.box_iou_rotated(a[i:i + 1], b[i:i + 1], version)
        got = _iou(_t(a[i:i + 1], cuda), _t(b[i:i + 1], cuda), version).cpu().numpy()
        assert np.isfinite(got).all(), (i, a[i], b[i], got)
        assert np.abs(got - want).max() <= IOU_TOL, (i, a[i], b[i], got, want)
    want0 = oracle_c.box_iou_rotated(a[:1], b[:1], 0)
    assert want0.view(np.int32)[0, 0] == np.float32(6.2644885e-06).view(np.int32)


def test_iou_s2anet_anchor_grid(cuda, oracle_c):
    """gt x the exact 21 824-anchor S2ANet grid (SURVEY 8d micro-bench shape, K=16)."""
    from rs_detection_amd import ops
    rng = np.random.default_rng(7)
    gts, anchors = dota_boxes(rng, 16), s2anet_anchors()
    assert anchors.shape == (21824, 5)
    want = oracle_c.box_iou_rotated(gts, anchors, 0)
    got = ops.box_iou_rotated(_t(gts, cuda), _t(anchors, cuda)).cpu().numpy()
    assert np.abs(got - want).max() <= IOU_TOL
    assert (got.view(np.int32) == want.view(np.int32)).mean() > 0.9999


def test_iou_empty_and_stride6(cuda, oracle_c):
    from rs_detection_amd import ops
    e = torch.zeros((0, 5), device=cuda)
    b = _t(dota_boxes(np.random.default_rng(0), 4, 100), cuda)
    assert ops.box_iou_rotated(e, b).shape == (0, 4)
    assert ops.box_iou_rotated(b, e).shape == (4, 0)
    # (n,6) rows with a trailing score column
    rng = np.random.default_rng(1)
    b6 = np.concatenate([dota_boxes(rng, 33, 150), rng.uniform(0, 1, (33, 1)).astype(np.float32)], 1)
    want = oracle_c.box_iou_rotated(b6[:, :5], b6[:, :5], 0)
    got = ops.box_iou_rotated(_t(b6, cuda), _t(b6, cuda)).cpu().numpy()
    assert np.abs(got - want).max() <= IOU_TOL


def test_iou_properties_full_size(cuda):
    """BASELINE full size (K=400 x A=21824) through size-independent properties:
    range, self-IoU = 1 on the diagonal, symmetry within tolerance, zero for far pairs."""
    from rs_detection_amd import ops
    rng = np.random.default_rng(3)
    g = _t(dota_boxes(rng, 400), cuda)
    a = _t(s2anet_anchors(), cuda)
    iou = ops.box_iou_rotated(g, a)
    assert iou.shape == (400, 21824)
    assert float(iou.min()) >= 0 and float(iou.max()) <= 1 + 1e-5
    iou_t = ops.box_iou_rotated(a, g)
    assert float((iou - iou_t.t()).abs().max()) <= IOU_TOL
    self_iou = ops.box_iou_rotated(g, g)
    assert float((self_iou.diagonal() - 1).abs().max()) <= 1e-5
    far = g.clone()
    far[:, 0] += 1e5
    assert float(ops.box_iou_rotated(g, far).abs().max()) == 0.0


def test_iou_work_queue_overflow_is_exact(cuda, oracle_c):
    """More overlapping pairs than the work queue holds (4 Mi): 2 200 x 2 200 boxes from one tight pile, every pair
    overlaps, so ~0.6 M pairs overflow their queue shard and are clipped by the filter workgroups themselves.
    Every entry must still be the exact IoU: sampled rows against the oracle, the diagonal == 1."""
    from rs_detection_amd import ops
    rng = np.random.default_rng(21)
    n = 2200
    b = np.tile(np.array([[500, 500, 120, 60, 0.4]], np.float32), (n, 1))
    b[:, :2] += rng.normal(0, 3, (n, 2)).astype(np.float32)
    b[:, 2:4] *= np.exp(rng.normal(0, 0.05, (n, 2))).astype(np.float32)
    b[:, 4] += rng.normal(0, 0.05, n).astype(np.float32)
    got = ops.box_iou_rotated(_t(b, cuda), _t(b, cuda)).cpu().numpy()
    assert (got > 0).all() and np.abs(np.diag(got) - 1).max() <= 1e-5
    rows = rng.choice(n, 12, replace=False)
    want = oracle_c.box_iou_rotated(b[rows], b, 0)
    assert np.abs(got[rows] - want).max() <= IOU_TOL
    assert (got[rows].view(np.int32) == want.view(np.int32)).mean() > 0.999


def _clustered(rng, n, with_label):
    centres = dota_boxes(rng, max(n // 10, 1), 600)
    idx = rng.integers(0, centres.shape[0], n)
    d = centres[idx].copy()
    d[:, :2] += rng.normal(0, 3, (n, 2)).astype(np.float32)
    d[:, 4] += rng.normal(0, 0.1, n).astype(np.float32)
    scores = rng.uniform(0.05, 1, n).astype(np.float32)
    if with_label:
        d = np.concatenate([d, rng.integers(0, 15, (n, 1)).astype(np.float32)], 1)
    return d, scores


@pytest.mark.parametrize("box_len", [5, 6])
@pytest.mark.parametrize("n,thr", [(1, 0.1), (3, 0.3), (64, 0.1), (65, 0.5), (700, 0.1), (2000, 0.1), (2000, 0.8)])
def test_nms_vs_oracle_bit_exact(cuda, oracle_c, box_len, n, thr):
    from rs_detection_amd.ops import nms_rotated_keep_mask
    rng = np.random.default_rng(n * 7 + box_len)
    d, s = _clustered(rng, n, box_len == 6)
    order = np.argsort(-s, kind="stable").astype(np.int32)
    want = oracle_c.nms_rotated(d, order, thr)
    got = nms_rotated_keep_mask(_t(d, cuda), _t(order, cuda), thr, box_len).cpu().numpy()
    assert (got == want).all(), (int((got != want).sum()), n)


def test_nms_suppression_chains_and_dense_blocks(cuda, oracle_c):
    """Shapes that stress the device sweep rather than the clipper:
    (a) a chain -- box i overlaps only i+1 -- where the greedy answer alternates and the in-block
        fixpoint needs one round per box (64 rounds per block, decisions crossing block borders);
    (b) one dense pile: every box suppresses every later one, so a block's entry list is far longer
        than the sweep's prefetch window and most blocks arrive already fully removed;
    (c) chain and pile mixed under a random score order."""
    from rs_detection_amd.ops import nms_rotated_keep_mask
    n = 333
    chain = np.zeros((n, 5), np.float32)
    chain[:, 0] = np.arange(n) * 6.0   # 10-wide boxes, 6 apart: IoU(i,i+1)=0.25, IoU(i,i+2)=0
    chain[:, 1] = 50
    chain[:, 2], chain[:, 3] = 10, 10
    order = np.arange(n, dtype=np.int32)                      # scores descending along the chain
    want = oracle_c.nms_rotated(chain, order, 0.2)
    assert want[::2].all() and not want[1::2].any()            # the oracle itself alternates
    got = nms_rotated_keep_mask(_t(chain, cuda), _t(order, cuda), 0.2, 5).cpu().numpy()
    assert (got == want).all()

    rng = np.random.default_rng(5)
    pile = np.tile(np.array([[300, 300, 80, 40, 0.3]], np.float32), (3000, 1))
    pile[:, :2] += rng.normal(0, 1.0, (3000, 2)).astype(np.float32)
    order = rng.permutation(3000).astype(np.int32)
    want = oracle_c.nms_rotated(pile, order, 0.3)
    got = nms_rotated_keep_mask(_t(pile, cuda), _t(order, cuda), 0.3, 5).cpu().numpy()
    assert (got == want).all() and want.sum() == 1

    both = np.concatenate([chain, pile[:1500]])
    order = rng.permutation(both.shape[0]).astype(np.int32)
    want = oracle_c.nms_rotated(both, order, 0.2)
    got = nms_rotated_keep_mask(_t(both, cuda), _t(order, cuda), 0.2, 5).cpu().numpy()
    assert (got == want).all(), int((got != want).sum())


def test_nms_two_tier_decisions_at_the_threshold(cuda, oracle_c):
    """The mask kernel settles a pair from the Green-integral IoU only when that sits further than the tier budget from
    the threshold and no corner lies on an edge (csrc/nms_rotated.hip, rsdet_geom_fast.h); everything else goes through
    the reference-order clipper.  Families built to sit ON the threshold: integer boxes (IoU exactly 1/4, 1/3, 1/2 for
    thousands of pairs, and the reference's hull scan dropping a vertex on shared edges), 3-4-5 rotations, and
    near-copies of the same box (IoU ~ 1, corners within 1e-5 .. 1e-2 px of the other box's edges)."""
    from rs_detection_amd.ops import nms_rotated_keep_mask
    rng = np.random.default_rng(77)
    n = 1500
    ib = np.stack([rng.integers(0, 60, n), rng.integers(0, 60, n), rng.integers(1, 12, n) * 2,
                   rng.integers(1, 12, n) * 2, rng.choice([0, np.pi / 2, np.pi, -np.pi / 2], n)], 1).astype(np.float32)
    ib345 = ib.copy()
    ang = np.arctan2(3, 4)
    ib345[:, 4] = rng.choice([ang, -ang, ang + np.pi / 2, 0], n)
    base = dota_boxes(rng, 300, 400)
    near = np.concatenate([base] + [(base + rng.normal(0, e, base.shape)).astype(np.float32)
                                    for e in (1e-5, 1e-3, 1e-2, 0.3)])
    n_dec = 0
    for name, d, thrs in (("integer", ib, (0.25, 1 / 3, 0.5, 0.2, 0.1, 0.0)),
                          ("3-4-5", ib345, (0.25, 1 / 3, 0.5, 0.1)),
                          ("near copies", near, (0.5, 0.9, 0.99, 0.999, 1.0))):
        order = rng.permutation(d.shape[0]).astype(np.int32)
        for thr in thrs:
            want = oracle_c.nms_rotated(d, order, np.float32(thr))
            got = nms_rotated_keep_mask(_t(d, cuda), _t(order, cuda), float(np.float32(thr)), 5).cpu().numpy()
            assert (got == want).all(), (name, thr, int((got != want).sum()))
            n_dec += int(want.sum())
    assert n_dec > 1000


@pytest.mark.parametrize("n,n_labels", [(700, 3), (3000, 15), (3000, 200), (500, 1), (130, 130)])
def test_nms_label_major_order_and_segmented_sweep(cuda, oracle_c, n, n_labels):
    """Class-aware NMS with the label-major order (tiles of disjoint label ranges skipped, one concurrent sweep per
    label run; 200 labels: more runs than sweep workgroups; runs sharing 64-box blocks) == the oracle's greedy loop
    in plain score order."""
    from rs_detection_amd.ops import nms_rotated_keep_mask, ml_nms_rotated
    from rs_detection_amd.ops.nms_rotated import _label_major_order
    rng = np.random.default_rng(n + n_labels)
    d, s = _clustered(rng, n, False)
    labels = rng.integers(0, n_labels, n)
    d6 = np.concatenate([d, labels[:, None].astype(np.float32)], 1)
    want = oracle_c.nms_rotated(d6, np.argsort(-s, kind="stable").astype(np.int32), 0.1)
    lorder = _label_major_order(_t(s, cuda), torch.from_numpy(labels).to(cuda))
    lab_sorted = labels[lorder.cpu().numpy()]
    assert (np.diff(lab_sorted) >= 0).all()
    got = nms_rotated_keep_mask(_t(d6, cuda), lorder, 0.1, 6, label_major=True).cpu().numpy()
    assert (got == want).all(), int((got != want).sum())
    got2 = nms_rotated_keep_mask(_t(d6, cuda), lorder, 0.1, 6, label_major=False).cpu().numpy()   # same order, one sweep
    assert (got2 == want).all()
    idx = ml_nms_rotated(_t(d, cuda), _t(s, cuda), torch.from_numpy(labels).to(cuda), 0.1).cpu().numpy()
    assert (idx == np.nonzero(want)[0]).all()


def test_nms_known_answer_and_api(cuda):
    from rs_detection_amd.ops import nms_rotated, ml_nms_rotated
    dets = _t(np.array([[0, 0, 1, 1, 0], [0, 0, .5, .5, .3], [0, 0, .9, .9, 0]], np.float32), cuda)
    scores = _t(np.array([.1, .2, .3], np.float32), cuda)
    labels = torch.tensor([1, 1, 1], device=cuda)
    assert nms_rotated(dets, scores, 0.3).tolist() == [2]            # nms_rotated.py:598-603
    assert ml_nms_rotated(dets, scores, labels, 0.3).tolist() == [2]
    assert nms_rotated(torch.zeros((0, 5), device=cuda), torch.zeros((0,), device=cuda), 0.3).numel() == 0


def test_nms_idempotent_full_size(cuda):
    """20 000 clustered boxes: NMS of the kept set keeps everything; kept set is sorted ascending."""
    from rs_detection_amd.ops import nms_rotated
    rng = np.random.default_rng(11)
    d, s = _clustered(rng, 20000, False)
    dt, st = _t(d, cuda), _t(s, cuda)
    keep = nms_rotated(dt, st, 0.1)
    assert (keep[1:] > keep[:-1]).all()
    keep2 = nms_rotated(dt[keep], st[keep], 0.1)
    assert keep2.numel() == keep.numel()


def test_multiclass_nms_vs_oracle(cuda, oracle_c):
    from rs_detection_amd.ops import multiclass_nms_rotated
    rng = np.random.default_rng(5)
    n, ncls = 600, 15
    boxes, _ = _clustered(rng, n, False)
    scores = np.zeros((n, ncls + 1), np.float32)
    scores[:, 1:] = rng.uniform(0, 0.2, (n, ncls)) * (rng.uniform(0, 1, (n, ncls)) > 0.8)
    dets, labels = multiclass_nms_rotated(_t(boxes, cuda), _t(scores, cuda), 0.05, dict(type='nms_rotated', iou_thr=0.1), 2000)
    # oracle restatement of nms_rotated.py:563-596
    valid = scores[:, 1:] > 0.05
    ii, cc = np.nonzero(valid)
    bb, ss = boxes[ii], scores[:, 1:][valid]
    d6 = np.concatenate([bb, cc[:, None].astype(np.float32)], 1)
    order = np.argsort(-ss, kind="stable").astype(np.int32)
    keep = np.nonzero(oracle_c.nms_rotated(d6, order, 0.1))[0]
    bb, ss, cc = bb[keep], ss[keep], cc[keep]
    inds = np.argsort(-ss, kind="stable")[:2000]
    np.testing.assert_allclose(dets.cpu().numpy(), np.concatenate([bb[inds], ss[inds, None]], 1), atol=1e-6)
    assert (labels.cpu().numpy() == cc[inds]).all()


@pytest.mark.parametrize("ks", [(16, 100, 1, 40), (5,), (400, 3)])
def test_grouped_iou_and_assign_bit_exact(cuda, oracle_c, ks):
    """Batched IoU + MaxIoUAssigner: gt_inds must be bit-exact vs the oracle (north_star)."""
    from rs_detection_amd import ops
    rng = np.random.default_rng(sum(ks))
    anchors = s2anet_anchors()
    gts = [dota_boxes(rng, k) for k in ks]
    labels = [rng.integers(1, 16, k).astype(np.int32) for k in ks]
    offs = np.concatenate([[0], np.cumsum(ks)]).astype(np.int32)
    b1 = _t(np.concatenate(gts), cuda)
    ro = _t(offs, cuda)
    ov = ops.box_iou_rotated_grouped(b1, ro, max(ks), _t(anchors, cuda))
    gi, mo, lb = ops.assign_wrt_overlaps(ov, ro, max(ks), 0.5, 0.4, 0.0, True, True, _t(np.concatenate(labels), cuda), 0)
    ov_np = ov.cpu().numpy()
    for g, k in enumerate(ks):
        want_ov = oracle_c.box_iou_rotated(gts[g], anchors, 0)
        got_ov = ov_np[offs[g]:offs[g + 1]]
        assert np.abs(got_ov - want_ov).max() <= IOU_TOL
        # assignment oracle on the oracle's own IoUs and on the GPU IoUs
        w_gi, w_mo, w_lb = oracle_c.assign_wrt_overlaps(got_ov, 0.5, 0.4, 0.0, True, True, labels[g], 0)
        assert (gi[g].cpu().numpy() == w_gi).all()
        assert (mo[g].cpu().numpy() == w_mo).all()
        assert (lb[g].cpu().numpy() == w_lb).all()
        o_gi, _, _ = oracle_c.assign_wrt_overlaps(want_ov, 0.5, 0.4, 0.0, True, True, labels[g], 0)
        assert (gi[g].cpu().numpy() == o_gi).all(), "anchor indices differ from the CPU path"


def test_grouped_iou_per_image_columns(cuda, oracle_c):
    """ODM case: every image has its own (refined) anchor set."""
    from rs_detection_amd import ops
    rng = np.random.default_rng(9)
    ks = (12, 30)
    A = 1500
    gts = [dota_boxes(rng, k, 300) for k in ks]
    cols = np.stack([dota_boxes(rng, A, 300) for _ in ks])
    offs = np.concatenate([[0], np.cumsum(ks)]).astype(np.int32)
    ov = ops.box_iou_rotated_grouped(_t(np.concatenate(gts), cuda), _t(offs, cuda), max(ks), _t(cols, cuda)).cpu().numpy()
    for g in range(2):
        want = oracle_c.box_iou_rotated(gts[g], cols[g], 0)
        assert np.abs(ov[offs[g]:offs[g + 1]] - want).max() <= IOU_TOL
