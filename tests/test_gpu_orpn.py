"""csrc/orpn.hip against the tensor-operation forms it replaces (and NumPy restatements of their selection rules):
RandomSampler.sample_masked (models/boxes/sampler.py:57-180 of the reference), MidpointOffsetCoder.decode
(models/boxes/coder.py:372-433), obb2hbb (ops/bbox_transforms.py:572-578) and the proposal routine of the Oriented RPN
(models/roi_heads/oriented_rpn_head.py:135-222)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.fixture
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


def _np_sample(gt, pri, num, num_pos, ub):
    """The selection rule in NumPy: the num_pos positives with the largest draws (ties: lower index), then negatives up to
    `num` (neg_pos_ub honoured); both lists ascending."""
    n = gt.shape[0]
    idx = np.arange(n)
    order = np.lexsort((idx, -pri))                     # by draw descending, index ascending
    pos = order[gt[order] > 0][:min(num_pos, n)]
    quota = num - len(pos)
    if ub >= 0:
        quota = min(quota, int(np.float32(ub) * np.float32(max(len(pos), 1))))
    neg = order[gt[order] == 0][:max(quota, 0)]
    return np.sort(pos), np.sort(neg)


@pytest.mark.parametrize("case", ["rpn", "roi_gt", "few_pos", "no_pos", "ub", "ties", "all_equal", "tiny", "f64"])
def test_sample_masked_selects_the_reference_sets(cuda, case):
    from rs_detection_amd.ops import orpn
    rng = np.random.default_rng(sum(map(ord, case)))
    n, k_gt, num, num_pos, ub, f64 = 300_000, 0, 256, 128, -1.0, False
    p_pos, p_neg = 0.002, 0.9
    if case == "roi_gt":
        n, k_gt, num, num_pos, p_pos, p_neg = 2000, 23, 512, 128, 0.2, 0.7
    elif case == "few_pos":
        p_pos = 0.0001
    elif case == "no_pos":
        p_pos = 0.0
    elif case == "ub":
        ub, p_pos = 3.0, 0.00005
    elif case == "tiny":
        n, num, num_pos = 100, 256, 128
    elif case == "f64":
        f64 = True
    gt = np.where(rng.random(n) < p_pos, rng.integers(1, 9, n), np.where(rng.random(n) < p_neg, 0, -1)).astype(np.int32)
    valid = rng.random(n) < 0.8 if case == "roi_gt" else None
    pri = rng.random(n + k_gt)
    if case == "ties":
        pri = np.floor(pri * 40) / 40                    # 40 distinct values: every threshold is a tie
    if case == "all_equal":
        pri[:] = 0.5                                     # the tie list overflows: the ordered-scan fallback
    pri = pri if f64 else pri.astype(np.float32)
    ext = np.concatenate([np.arange(1, k_gt + 1, dtype=np.int32), gt if valid is None else np.where(valid, gt, -1)])
    want_pos, want_neg = _np_sample(ext, pri.astype(np.float64), num, num_pos, ub)
    t = lambda a: None if a is None else torch.from_numpy(a).to(cuda)
    assert orpn.sampler_applies(t(gt), t(pri), num)
    inds, is_pos, val, assigned, counts = orpn.sample_masked(t(gt), t(valid), k_gt, t(pri), num, num_pos, ub)
    inds, is_pos, val, assigned, counts = [x.cpu().numpy() for x in (inds, is_pos, val, assigned, counts)]
    assert counts.tolist() == [len(want_pos), len(want_neg)]
    npos, nneg = len(want_pos), len(want_neg)
    np.testing.assert_array_equal(inds[:npos], want_pos)
    np.testing.assert_array_equal(inds[npos:npos + nneg], want_neg)
    assert is_pos[:npos].all() and not is_pos[npos:].any()
    assert val[:npos + nneg].all() and not val[npos + nneg:].any()
    assert (inds[npos + nneg:] == 0).all()
    np.testing.assert_array_equal(assigned[:npos], ext[want_pos] - 1)
    assert (assigned[npos:] == 0).all()


def test_sampler_route_equals_the_tensor_route(cuda):
    """BaseSampler.sample_masked through the kernel == through the tensor operations (distinct draws: torch.topk's tie
    order is unspecified), every field of MaskedSamples on the used slots."""
    from rs_detection_amd.models.boxes.assigner import AssignResult
    from rs_detection_amd.models.boxes.sampler import RandomSamplerRotated
    from rs_detection_amd.ops import orpn
    rng = np.random.default_rng(3)
    n, K = 2000, 17
    pri = torch.from_numpy(rng.permutation(1 << 12)[:n + K].astype(np.float32) / (1 << 12)).to(cuda)
    gt_inds = torch.from_numpy(np.where(rng.random(n) < 0.1, rng.integers(1, K + 1, n), np.where(rng.random(n) < 0.8, 0, -1))
                               .astype(np.int32)).to(cuda)
    labels = torch.from_numpy(rng.integers(0, 15, n).astype(np.int32)).to(cuda)
    gt_labels = torch.from_numpy(rng.integers(0, 15, K)).to(cuda)
    boxes = torch.from_numpy(rng.random((n, 6)).astype(np.float32) * 100).to(cuda)
    gts = torch.from_numpy(rng.random((K, 5)).astype(np.float32) * 100).to(cuda)
    valid = torch.from_numpy(rng.random(n) < 0.9).to(cuda)
    sampler = RandomSamplerRotated(num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True)
    sampler.priorities = lambda m, dev: pri[:m]
    out = []
    for on in (True, False):
        orpn._ON = on
        try:
            out.append(sampler.sample_masked(AssignResult(K, gt_inds, None, labels), boxes, gts, gt_labels, valid=valid))
        finally:
            orpn._ON = True
    a, b = out
    used = b.valid.cpu().numpy()
    assert int(a.n_pos) == int(b.n_pos) and int(a.n_neg) == int(b.n_neg) and used.sum() == int(b.n_pos) + int(b.n_neg)
    np.testing.assert_array_equal(a.valid.cpu().numpy(), used)
    np.testing.assert_array_equal(a.is_pos.cpu().numpy(), b.is_pos.cpu().numpy())
    pos = b.is_pos.cpu().numpy()
    for f in ("inds", "bboxes", "pos_gt_labels"):
        np.testing.assert_array_equal(getattr(a, f).cpu().numpy()[used], getattr(b, f).cpu().numpy()[used])
    np.testing.assert_array_equal(a.pos_gt_bboxes.cpu().numpy()[pos], b.pos_gt_bboxes.cpu().numpy()[pos])


def _rand_anchors_deltas(rng, n):
    c = rng.uniform(50, 950, (n, 2))
    wh = rng.uniform(8, 300, (n, 2))
    anchors = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    deltas = (rng.normal(0, 0.4, (n, 6)) * [1, 1, 1, 1, 1.5, 1.5]).astype(np.float32)
    return anchors, deltas


def test_midpoint_decode_and_obb2hbb_equal_the_tensor_forms(cuda):
    from rs_detection_amd.models.boxes.coder import MidpointOffsetCoder
    from rs_detection_amd.ops import orpn
    from rs_detection_amd.ops.bbox_transforms import obb2hbb, obb2poly
    rng = np.random.default_rng(0)
    anchors, deltas = _rand_anchors_deltas(rng, 20_000)
    a, d = torch.from_numpy(anchors).to(cuda), torch.from_numpy(deltas).to(cuda)
    coder = MidpointOffsetCoder(target_stds=(1., 1., 1., 1., 0.5, 0.5))
    got = coder.decode(a, d)
    orpn._ON = False
    try:
        want = coder.decode(a, d)
    finally:
        orpn._ON = True
    assert got.shape == want.shape == (20_000, 5)
    # as polygons (a square's w/h order and angle are one rounding away from their alternatives)
    pg, pw = obb2poly(got).cpu().numpy().reshape(-1, 4, 2), obb2poly(want).cpu().numpy().reshape(-1, 4, 2)
    err = np.min([np.abs(np.roll(pg, s, 1) - pw).max((1, 2)) for s in range(4)], 0)
    assert np.quantile(err, 0.999) < 2e-3 and (err < 0.05).all(), (np.quantile(err, 0.999), err.max())
    same = (np.abs(got.cpu().numpy() - want.cpu().numpy()).max(1) < 1e-3).mean()
    assert same > 0.995, same
    g = got.cpu().numpy()
    assert (g[:, 2] >= g[:, 3]).all() and (g[:, 4] >= -np.pi / 2 - 1e-6).all() and (g[:, 4] < np.pi / 2 + 1e-6).all()
    h = obb2hbb(got)
    orpn._ON = False
    try:
        hw = obb2hbb(got)
    finally:
        orpn._ON = True
    np.testing.assert_allclose(h.cpu().numpy(), hw.cpu().numpy(), rtol=1e-5, atol=2e-4)


def _rpn(cuda, nms_pre=2000, nms_post=2000, min_bbox_size=0):
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.config import Config
    from rs_detection_amd.utils.registry import HEADS, build_from_cfg
    cfg = Config(os.path.join(ROOT, "configs", "orcnn", "orcnn_van3_7_anchor.py")).dump()["model"]["rpn"]
    cfg.update(nms_pre=nms_pre, nms_post=nms_post, min_bbox_size=min_bbox_size)
    torch.manual_seed(0)
    return build_from_cfg(cfg, HEADS).to(cuda).train()


def _maps(rng, B, size, A, equal=False):
    sizes = [(size // s, size // s) for s in (4, 8, 16, 32, 64)]
    cls = [rng.normal(-2, 1.5, (B, A, h, w)).astype(np.float32) for h, w in sizes]
    if equal:
        cls = [np.zeros_like(c) for c in cls]
    reg = [rng.normal(0, 0.3, (B, 6 * A, h, w)).astype(np.float32) for h, w in sizes]
    return cls, reg


@pytest.mark.parametrize("case", ["1024", "small_pre", "min_size", "equal_scores", "quantised"])
def test_batched_proposals_equal_the_per_image_routine(cuda, case):
    """rsdet_orpn_proposals_f32 == _get_bboxes_single(fixed=True) image by image (whose decode / obb2hbb run the same device
    functions): the same rows in the same order, zero rows behind them."""
    from rs_detection_amd.ops import orpn
    rng = np.random.default_rng(11)
    size, B, kw = 1024, 2, {}
    if case == "small_pre":
        size, kw = 256, dict(nms_pre=300, nms_post=100)
    elif case == "min_size":
        size, kw = 512, dict(min_bbox_size=40)
    elif case in ("equal_scores", "quantised"):
        size = 512
    rpn = _rpn(cuda, **kw)
    cls, reg = _maps(rng, B, size, 7, equal=case == "equal_scores")
    if case == "quantised":
        cls = [np.round(c * 2) / 2 for c in cls]                 # a few distinct scores: every level's threshold is a tie
    t = lambda xs: [torch.from_numpy(x).to(cuda) for x in xs]
    cls, reg = t(cls), t(reg)
    targets = [dict(img_size=(size, size), pad_shape=(size, size))] * B
    with torch.no_grad():
        got = rpn.get_bboxes(cls, reg, targets, fixed=True)
        sizes = [tuple(c.shape[-2:]) for c in cls]
        anchors = rpn.anchor_generator.grid_anchors(sizes, device=cuda)
        for i in range(B):
            want, wreal = rpn._get_bboxes_single([c[i] for c in cls], [r[i] for r in reg], anchors, (size, size), fixed=True)
            d, real = got[i]
            assert d.shape == want.shape == (rpn.nms_post, 6)
            assert torch.equal(real, wreal), (int(real.sum()), int(wreal.sum()))
            assert torch.equal(d, want)
            assert int(real.sum()) > 10


def test_batched_proposals_close_to_the_tensor_operations(cuda):
    """... and against the routine built from tensor operations only (the round-5 form): the same proposals up to the
    arithmetic of the decode (a library matmul inside rectpoly2obb; NMS decisions within rounding of the threshold may
    differ: < 0.5 % of the rows)."""
    from rs_detection_amd.ops import orpn
    rng = np.random.default_rng(5)
    size, B = 512, 2
    rpn = _rpn(cuda)
    cls, reg = _maps(rng, B, size, 7)
    t = lambda xs: [torch.from_numpy(x).to(cuda) for x in xs]
    cls, reg = t(cls), t(reg)
    targets = [dict(img_size=(size, size), pad_shape=(size, size))] * B
    with torch.no_grad():
        got = rpn.get_bboxes(cls, reg, targets, fixed=True)
        orpn._ON = False
        try:
            want = rpn.get_bboxes(cls, reg, targets, fixed=True)
        finally:
            orpn._ON = True
    for (d, real), (w, wreal) in zip(got, want):
        n, m = int(real.sum()), int(wreal.sum())
        assert abs(n - m) <= max(2, m // 200), (n, m)
        sg, sw = d[:n, 5].cpu().numpy(), w[:m, 5].cpu().numpy()
        k = min(n, m)
        assert (sg[:k] == sw[:k]).mean() > 0.99                      # same scores in the same order
        rows = np.nonzero(sg[:k] == sw[:k])[0]
        err = np.abs(d[:k].cpu().numpy()[rows, :4] - w[:k].cpu().numpy()[rows, :4]).max(1)
        assert np.quantile(err, 0.99) < 5e-3, np.quantile(err, 0.99)


def test_rpn_losses_on_the_samples_equal_the_dense_target_maps(cuda, monkeypatch):
    """OrientedRPNHead.loss through rsdet_orpn_loss (the sampled anchors only) == through the dense target maps of
    get_targets_masked + loss_single per level: the ten scalars and the gradients of all ten prediction maps."""
    from rs_detection_amd.models.boxes.sampler import RandomSampler
    from rs_detection_amd.ops import orpn
    from rs_detection_amd.utils import synthetic as syn
    rng = np.random.default_rng(2)
    size, B = 512, 2
    rpn = _rpn(cuda)
    cls, reg = _maps(rng, B, size, 7)
    pri = torch.from_numpy(np.random.default_rng(9).random(1 << 20).astype(np.float32)).to(cuda)
    monkeypatch.setattr(RandomSampler, "priorities", staticmethod(lambda n, dev: pri[:n]))
    targets = []
    for t in syn.synthetic_targets(B, img=size, num_classes=10):
        t = dict(t, rboxes=torch.from_numpy(t["rboxes"][:20]).to(cuda), rboxes_ignore=None, img_size=(size, size),
                 pad_shape=(size, size))
        targets.append(t)
    res = []
    for on in (True, False):
        c = [torch.from_numpy(x).to(cuda).requires_grad_(True) for x in cls]
        r = [torch.from_numpy(x).to(cuda).requires_grad_(True) for x in reg]
        orpn._ON = on
        try:
            losses = rpn.loss(c, r, targets)
            w = torch.linspace(0.5, 1.5, 10, device=cuda)                 # a different upstream gradient per scalar
            total = sum(wi * li for wi, li in zip(w, losses["loss_rpn_cls"] + losses["loss_rpn_bbox"]))
            total.backward()
        finally:
            orpn._ON = True
        res.append((losses, [x.grad for x in c + r]))
    (la, ga), (lb, gb) = res
    assert sum(float(x) for x in lb["loss_rpn_bbox"]) > 0
    for k in ("loss_rpn_cls", "loss_rpn_bbox"):
        assert len(la[k]) == len(lb[k]) == 5
        for x, y in zip(la[k], lb[k]):
            assert abs(float(x) - float(y)) <= 2e-5 * max(abs(float(y)), 1e-3) + 1e-7, (k, float(x), float(y))
    for x, y in zip(ga, gb):
        assert x.shape == y.shape
        assert int((x != 0).sum()) == int((y != 0).sum())
        assert torch.allclose(x, y, rtol=1e-4, atol=1e-9), float((x - y).abs().max())


def test_roi_head_targets_equal_the_tensor_route(cuda, monkeypatch):
    """OrientedHead._forward_train_masked with the samples -> RoIs / labels / encoded targets as one launch per image ==
    the same through tensor operations: both losses and the gradient reaching the pyramid."""
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.config import Config
    from rs_detection_amd.models.boxes.sampler import RandomSampler
    from rs_detection_amd.ops import orpn
    from rs_detection_amd.utils.registry import HEADS, build_from_cfg
    from rs_detection_amd.utils import synthetic as syn
    cfg = Config(os.path.join(ROOT, "configs", "orcnn", "orcnn_van3_7_anchor.py")).dump()["model"]["bbox_head"]
    torch.manual_seed(0)
    head = build_from_cfg(cfg, HEADS).to(cuda).train()
    rng = np.random.default_rng(4)
    size, B, P = 512, 2, 2000
    pri = torch.from_numpy(np.random.default_rng(1).permutation(1 << 14).astype(np.float32) / (1 << 14)).to(cuda)
    monkeypatch.setattr(RandomSampler, "priorities", staticmethod(lambda n, dev: pri[:n]))
    targets, props = [], []
    for t in syn.synthetic_targets(B, img=size, num_classes=10):
        rb = t["rboxes"][:30]
        targets.append(dict(rboxes=torch.from_numpy(rb).to(cuda), labels=torch.from_numpy(t["labels"][:30]).to(cuda)))
        near = rb[rng.integers(0, len(rb), P // 2)].copy()
        near[:, 4] *= -1
        near[:, :2] += rng.normal(0, 3, (P // 2, 2))
        near[:, 2:4] *= np.exp(rng.normal(0, 0.1, (P // 2, 2)))
        far = np.concatenate([rng.uniform(20, size - 20, (P - P // 2, 2)), rng.uniform(10, 90, (P - P // 2, 2)),
                              rng.uniform(-1.5, 1.5, (P - P // 2, 1))], 1)
        d = np.concatenate([np.concatenate([near, far]), rng.random((P, 1))], 1).astype(np.float32)
        real = rng.random(P) < 0.9
        props.append((torch.from_numpy(d).to(cuda), torch.from_numpy(real).to(cuda)))
    res = []
    for on in (True, False):
        feats = [torch.from_numpy(np.random.default_rng(8 + l).normal(0, 1, (B, 256, size // s, size // s)).astype(np.float32))
                 .to(cuda).requires_grad_(True) for l, s in enumerate((4, 8, 16, 32, 64))]
        orpn._ON = on
        try:
            out = head._forward_train_masked(feats, props, targets)
            (out["loss_cls"] + 2.0 * out["orcnn_bbox_loss"]).sum().backward()
        finally:
            orpn._ON = True
        res.append((out, feats[0].grad))
    (a, ga), (b, gb) = res
    assert float(b["orcnn_bbox_loss"].sum()) > 0
    for k in ("loss_cls", "orcnn_bbox_loss"):
        assert torch.allclose(a[k], b[k], rtol=1e-5, atol=1e-7), (k, a[k], b[k])
    assert torch.allclose(ga, gb, rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("K,A,low,allm,minpos", [(25, 611072, True, True, 0.3), (400, 50000, True, True, 0.3),
                                                  (1, 3000, True, False, 0.3), (130, 20000, False, True, 0.3),
                                                  (700, 9000, True, True, 0.3), (60, 30000, True, False, 0.0),
                                                  (60, 30000, True, True, 0.0)])
def test_hbb_assignment_without_the_matrix_equals_the_matrix_route(cuda, K, A, low, allm, minpos):
    """rsdet_hbb_assign_f32 (two passes that recompute the horizontal IoU) == rsdet_bbox_overlaps_f32 +
    rsdet_assign_wrt_overlaps_f32 on the (K, A) matrix: gt_inds and max_overlaps bit for bit -- grid anchors with many exact
    ties (equal IoUs across anchors and across ground truths), duplicated ground truths, tiny boxes; with min_pos_iou = 0
    also a ground truth no anchor overlaps (row maximum 0 at the FIRST anchor, which the low-quality rule then assigns)."""
    from rs_detection_amd.models.boxes.assigner import MaxIoUAssigner
    from rs_detection_amd.ops import orpn
    rng = np.random.default_rng(K + A)
    side = int(np.sqrt(A / 7)) + 1
    cy, cx = np.meshgrid(np.arange(side) * 4.0 + 2, np.arange(side) * 4.0 + 2, indexing="ij")
    sizes = np.array([[16, 16], [32, 16], [16, 32], [64, 32], [32, 64], [23, 23], [90, 40]], np.float32)
    c = np.stack([cx, cy], -1).reshape(-1, 1, 2)
    anchors = np.concatenate([c - sizes[None] / 2, c + sizes[None] / 2], -1).reshape(-1, 4)[:A].astype(np.float32)
    ctr = rng.uniform(0, side * 4.0, (K, 2))
    wh = np.exp(rng.uniform(np.log(6), np.log(160), (K, 2)))
    gts = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    if K > 4:
        gts[3] = gts[1]                                   # a duplicated ground truth: equal rows
        gts[2] = anchors[len(anchors) // 2]               # an IoU of exactly 1
    if minpos == 0.0:
        gts[5] = np.array([-900, -900, -850, -870], np.float32)      # overlaps nothing
    a, g = torch.from_numpy(anchors).to(cuda), torch.from_numpy(gts).to(cuda)
    asg = MaxIoUAssigner(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=minpos, match_low_quality=low,
                         gt_max_assign_all=allm, ignore_iof_thr=-1)
    assert orpn.hbb_assign_applies(a, g)
    got = asg.assign(a, g, None, None)
    orpn._ON = False
    try:
        want = asg.assign(a, g, None, None)
    finally:
        orpn._ON = True
    assert got.gt_inds.dtype == want.gt_inds.dtype and torch.equal(got.gt_inds, want.gt_inds)
    assert torch.equal(got.max_overlaps, want.max_overlaps)
    assert int((got.gt_inds > 0).sum()) >= min(K, 3)
    lab = torch.from_numpy(rng.integers(0, 15, K)).to(cuda)
    gl = asg.assign(a, g, None, lab)
    orpn._ON = False
    try:
        wl = asg.assign(a, g, None, lab)
    finally:
        orpn._ON = True
    assert torch.equal(gl.labels, wl.labels)
