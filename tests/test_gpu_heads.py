"""GPU: head-level known answers.  The torch heads of the product, fed seeded prediction maps,
against NumPy restatements of the reference's host logic (oracle/heads.py: S2ANetHead.loss :322-508 -> 20 loss scalars,
get_bboxes_single :543-601 -> detections, OrientedRPNHead._get_bboxes_single :156-227 -> proposals), which in turn
stand on the per-op oracles (IoU / NMS pinned by the reference's own CPU source)."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import heads as H
from conftest import dota_boxes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STRIDES = [8, 16, 32, 64, 128]


def _head(cuda):
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.config import Config
    from rs_detection_amd.utils.registry import HEADS, build_from_cfg
    cfg = Config(os.path.join(ROOT, "configs", "s2anet", "s2anet_r50_fpn_1x_dota.py"))
    torch.manual_seed(0)
    return build_from_cfg(cfg.model["bbox_head"], HEADS).to(cuda), cfg.model["bbox_head"]


def _maps(rng, B, size, C):
    """Seeded prediction maps of a (size x size) input: per level (B, C|5, H, W) + refined anchors (B, H, W, 5)."""
    fam_cls, fam_box, odm_cls, odm_box, refined = [], [], [], [], []
    for s in STRIDES:
        f = -(-size // s)
        fam_cls.append(rng.normal(-3, 1.5, (B, C, f, f)).astype(np.float32))
        odm_cls.append(rng.normal(-3, 1.5, (B, C, f, f)).astype(np.float32))
        fam_box.append(rng.normal(0, 0.3, (B, 5, f, f)).astype(np.float32))
        odm_box.append(rng.normal(0, 0.3, (B, 5, f, f)).astype(np.float32))
        a = oracle.np_s2anet_grid_anchors((f, f), s)
        r = np.stack([a.copy() for _ in range(B)])
        r[:, :, :2] += rng.normal(0, 0.15 * 4 * s, (B, f * f, 2)).astype(np.float32)
        r[:, :, 2:4] *= np.exp(rng.normal(0, 0.25, (B, f * f, 2))).astype(np.float32)
        r[:, :, 4] += rng.normal(0, 0.4, (B, f * f)).astype(np.float32)
        refined.append(r.reshape(B, f, f, 5).astype(np.float32))
    return fam_cls, fam_box, refined, odm_cls, odm_box


@pytest.mark.parametrize("B,size,ks", [(2, 256, [12, 30]), (3, 128, [1, 7, 20])])
def test_s2anet_head_loss_known_answer(cuda, B, size, ks):
    head, hcfg = _head(cuda)
    head.train()
    rng = np.random.default_rng(B * 1000 + size)
    fam_cls, fam_box, refined, odm_cls, odm_box = _maps(rng, B, size, 15)
    gts = [dota_boxes(rng, k, float(size), 8, 120, 48) for k in ks]
    labs = [rng.integers(1, 16, k).astype(np.int32) for k in ks]
    t = lambda xs: [torch.from_numpy(x).to(cuda) for x in xs]
    metas = [dict(img_shape=(size, size), scale_factor=1.0, pad_shape=(size, size)) for _ in ks]
    got = head.loss(t(fam_cls), t(fam_box), t(refined), t(odm_cls), t(odm_box), t(gts), t(labs), metas)
    acfg = hcfg["train_cfg"]["fam_cfg"]["assigner"]
    stage = dict(pos_iou_thr=acfg["pos_iou_thr"], neg_iou_thr=acfg["neg_iou_thr"], min_pos_iou=acfg["min_pos_iou"],
                 means=hcfg["target_means"], stds=hcfg["target_stds"])
    want = H.np_s2anet_head_loss(fam_cls, fam_box, refined, odm_cls, odm_box, gts, labs, STRIDES, stage, stage,
                                 dict(gamma=2.0, alpha=0.25, loss_weight=1.0), dict(beta=1.0 / 9.0, loss_weight=1.0))
    assert set(got) == set(want) == {"loss_fam_cls", "loss_fam_bbox", "loss_odm_cls", "loss_odm_bbox"}
    n = 0
    for k in want:
        assert len(got[k]) == len(want[k]) == 5                      # one scalar per pyramid level
        for lvl in range(5):
            g, w = float(got[k][lvl]), float(want[k][lvl])
            assert abs(g - w) <= 2e-4 * max(abs(w), 1e-3) + 1e-6, (k, lvl, g, w)
            n += 1
    assert n == 20
    assert sum(float(x) for x in want["loss_fam_bbox"]) > 0 and sum(float(x) for x in want["loss_odm_bbox"]) > 0


def test_s2anet_get_bboxes_single_known_answer(cuda):
    head, hcfg = _head(cuda)
    head.eval()
    rng = np.random.default_rng(5)
    size = 256
    _, _, refined, odm_cls, odm_box = _maps(rng, 1, size, 15)
    for c in odm_cls:                                   # a few confident, well separated detections per level
        c -= 4.0
        idx = rng.integers(0, c[0, 0].size, max(c[0, 0].size // 40, 2))
        c.reshape(1, 15, -1)[0, rng.integers(0, 15, len(idx)), idx] = rng.uniform(1.0, 4.0, len(idx))
    cfg = dict(hcfg["test_cfg"])
    cfg["nms_pre"] = 300                                # exercises the per-level top-k at level 0 (1024 anchors)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(cuda)
    polys, scores, labels = head.get_bboxes_single([t(c[0]) for c in odm_cls], [t(b[0]) for b in odm_box],
                                                   [t(r[0].reshape(-1, 5)) for r in refined], (size, size), 2.0, cfg,
                                                   rescale=True)
    wp, ws, wl = H.np_s2anet_get_bboxes_single([c[0] for c in odm_cls], [b[0] for b in odm_box],
                                               [r[0].reshape(-1, 5) for r in refined], 2.0, cfg, hcfg["target_means"],
                                               hcfg["target_stds"], rescale=True)
    assert len(ws) > 10
    assert polys.shape[0] == len(ws), (polys.shape, len(ws))
    np.testing.assert_allclose(scores.cpu().numpy(), ws, rtol=1e-5, atol=1e-6)
    assert (labels.cpu().numpy() == wl).all()
    np.testing.assert_allclose(polys.cpu().numpy(), wp, rtol=1e-4, atol=2e-3)


def test_oriented_rpn_get_bboxes_single_known_answer(cuda):
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.config import Config
    from rs_detection_amd.utils.registry import HEADS, build_from_cfg
    cfg = Config(os.path.join(ROOT, "configs", "orcnn", "orcnn_van3_7_anchor.py")).dump()["model"]["rpn"]
    cfg.update(nms_pre=200, nms_post=150)
    torch.manual_seed(0)
    rpn = build_from_cfg(cfg, HEADS).to(cuda).eval()
    rng = np.random.default_rng(9)
    size, strides = 128, [4, 8, 16, 32, 64]
    sizes = [(-(-size // s), -(-size // s)) for s in strides]
    anchors = rpn.anchor_generator.grid_anchors(sizes, device=cuda)
    cls = [rng.normal(0, 2, (7, h, w)).astype(np.float32) for h, w in sizes]
    reg = [rng.normal(0, 0.4, (42, h, w)).astype(np.float32) for h, w in sizes]
    t = lambda x: torch.from_numpy(x).to(cuda)
    got = rpn._get_bboxes_single([t(c) for c in cls], [t(r) for r in reg], anchors, (size, size)).cpu().numpy()
    want = H.np_oriented_rpn_get_bboxes_single(cls, reg, [a.cpu().numpy() for a in anchors], cfg["bbox_coder"]["target_means"],
                                               cfg["bbox_coder"]["target_stds"], 200, 150, cfg["nms_thresh"], 0)
    assert got.shape == want.shape and got.shape[0] > 50 and got.shape[1] == 6
    np.testing.assert_allclose(got[:, 5], want[:, 5], rtol=1e-5, atol=1e-6)        # same proposals, same order
    np.testing.assert_allclose(got[:, :5], want[:, :5], rtol=1e-4, atol=5e-3)


# ---- Oriented R-CNN heads: known answers with a FIXED sampling choice --------------------------------
_PRIORITY = np.random.default_rng(7919).random(1 << 20)        # one fixed draw per candidate INDEX (float64: no ties)


def _fixed_choice_np(gallery, num):
    """stands in for gallery[randperm(len)[:num]] on both sides (jt.randperm is not reproducible across frameworks): the
    `num` candidates with the largest fixed draws -- a choice BOTH forms of the sampler can be handed: the reference-shaped
    one through ``random_choice(gallery, num)``, the fixed-size one (the train step's) through ``priorities(n)``."""
    return gallery[np.argsort(-_PRIORITY[gallery], kind="stable")[:num]]


@pytest.fixture
def fixed_choice(monkeypatch):
    from rs_detection_amd.models.boxes.sampler import RandomSampler
    table = {}

    def pri(dev):
        if dev not in table:
            table[dev] = torch.from_numpy(_PRIORITY).to(dev)
        return table[dev]

    def choice(gallery, num):
        return gallery[torch.topk(pri(gallery.device)[gallery], num)[1]]
    monkeypatch.setattr(RandomSampler, "random_choice", staticmethod(choice))
    monkeypatch.setattr(RandomSampler, "priorities", staticmethod(lambda n, dev: pri(dev)[:n]))
    return _fixed_choice_np


def _orcnn_cfg():
    from rs_detection_amd.config import Config
    return Config(os.path.join(ROOT, "configs", "orcnn", "orcnn_van3_7_anchor.py")).dump()["model"]


def test_oriented_rpn_loss_known_answer(cuda, fixed_choice):
    """OrientedRPNHead.loss (oriented_rpn_head.py:274-480): 2 x 5 per-level scalars + the per-image target maps
    against oracle/heads.np_oriented_rpn_loss on seeded maps, the reference's own config for the head."""
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.utils.registry import HEADS, build_from_cfg
    cfg = _orcnn_cfg()["rpn"]
    torch.manual_seed(0)
    rpn = build_from_cfg(cfg, HEADS).to(cuda).train()
    rng = np.random.default_rng(21)
    size, strides, B = 256, [4, 8, 16, 32, 64], 2
    sizes = [(size // s, size // s) for s in strides]
    cls = [rng.normal(-1, 1.5, (B, 7, h, w)).astype(np.float32) for h, w in sizes]
    reg = [rng.normal(0, 0.3, (B, 42, h, w)).astype(np.float32) for h, w in sizes]
    targets = []
    for k in (9, 25):
        rb = dota_boxes(rng, k, float(size), 12, 120, 60)
        rb[:, :2] = np.clip(rb[:, :2], 40, size - 40)
        targets.append(dict(rboxes=rb, rboxes_ignore=None, img_size=(size, size), pad_shape=(size, size)))
    t = lambda xs: [torch.from_numpy(x).to(cuda) for x in xs]
    tt = [dict(x, rboxes=torch.from_numpy(x["rboxes"]).to(cuda)) for x in targets]
    assert rpn.masked                                   # the train step's form: fixed-size samples, counts on the device
    got = rpn.loss(t(cls), t(reg), tt)
    rpn.masked = False                                  # ... and the reference-shaped index lists: the same losses
    got_lists = rpn.loss(t(cls), t(reg), tt)
    rpn.masked = True
    for k in got:
        for a, b in zip(got[k], got_lists[k]):
            assert abs(float(a) - float(b)) <= 1e-5 * max(abs(float(b)), 1e-3), (k, float(a), float(b))
    want, per = H.np_oriented_rpn_loss(cls, reg, targets, cfg, fixed_choice)
    assert sum(len(p[4]) for p in per) > 20 and all(len(p[5]) > 100 for p in per)       # the sampler had to choose
    for k in ("loss_rpn_cls", "loss_rpn_bbox"):
        assert len(got[k]) == len(want[k]) == 5
        for lvl in range(5):
            g, w = float(got[k][lvl]), float(want[k][lvl])
            assert abs(g - w) <= 2e-4 * max(abs(w), 1e-3) + 1e-6, (k, lvl, g, w)
    assert sum(want["loss_rpn_bbox"]) > 0
    # the target maps themselves, image by image (labels, weights, encoded targets)
    mla = rpn.anchor_generator.grid_anchors(sizes, device=cuda)
    for i, tg in enumerate(tt):
        vf = rpn.anchor_generator.valid_flags(sizes, tg["pad_shape"], device=cuda)
        lab, lw, bt, bw, pos, neg, _ = rpn._get_targets_single(mla, vf, tg)
        assert (lab.cpu().numpy() == per[i][0]).all() and (lw.cpu().numpy() == per[i][1]).all()
        assert (bw.cpu().numpy() == per[i][3]).all()
        np.testing.assert_allclose(bt.cpu().numpy(), per[i][2], rtol=1e-4, atol=1e-4)
        assert (pos.cpu().numpy() == per[i][4]).all() and (neg.cpu().numpy() == per[i][5]).all()
        mlab, mlw, mbt, mbw, npos, nneg = rpn._get_targets_single_masked(mla, vf, tg)      # the fixed-size form
        assert (mlab == lab).all() and (mlw == lw).all() and (mbw == bw).all() and torch.equal(mbt, bt)
        assert int(npos) == len(per[i][4]) and int(nneg) == len(per[i][5])


def test_oriented_head_known_answers(cuda, fixed_choice):
    """OrientedHead: assign + sample (oriented_head.py:566-588), get_bboxes_targets (:426-496), loss (:354-424) and
    get_bboxes (:498-536, :279-305) against oracle/heads.py on seeded proposals / predictions."""
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.utils.registry import HEADS, build_from_cfg
    cfg = _orcnn_cfg()["bbox_head"]
    torch.manual_seed(0)
    head = build_from_cfg(cfg, HEADS).to(cuda).train()
    rng = np.random.default_rng(33)
    size, ks, P = 512, (6, 40), 900
    gts, labs, props = [], [], []
    for k in ks:
        g = dota_boxes(rng, k, float(size), 16, 140, 60)
        g[:, 4] = rng.uniform(-np.pi / 2, np.pi / 2, k)
        near = g[rng.integers(0, k, P // 2)].copy()                    # jittered copies of the gts (theta sign as the
        near[:, 4] *= -1                                               # head sees them) + background boxes
        near[:, :2] += rng.normal(0, 4, (P // 2, 2))
        near[:, 2:4] *= np.exp(rng.normal(0, 0.15, (P // 2, 2)))
        near[:, 4] += rng.normal(0, 0.08, P // 2)
        far = dota_boxes(rng, P - P // 2, float(size), 16, 140, 60)
        p = np.concatenate([near, far]).astype(np.float32)
        p = np.concatenate([p, rng.uniform(0, 1, (P, 1)).astype(np.float32)], 1)[rng.permutation(P)]
        gts.append(g.astype(np.float32)), labs.append(rng.integers(1, 11, k).astype(np.int64)), props.append(p)
    samples, results = [], []
    for i in range(len(ks)):
        obb = torch.from_numpy(gts[i]).to(cuda).clone()
        obb[:, -1] *= -1
        lab0 = torch.from_numpy(labs[i]).to(cuda) - 1
        pr = torch.from_numpy(props[i]).to(cuda)
        ar = head.assigner.assign(pr, obb, None, lab0)
        res = head.sampler.sample(ar, pr, obb, lab0)
        want = H.np_oriented_head_sample(props[i], gts[i], labs[i], cfg, fixed_choice)
        assert (res.pos_inds.cpu().numpy() == want["pos_inds"]).all() and len(want["pos_inds"]) >= min(ks[i], 128)
        assert (res.neg_inds.cpu().numpy() == want["neg_inds"]).all()
        assert len(want["pos_inds"]) + len(want["neg_inds"]) == 512
        np.testing.assert_array_equal(res.pos_bboxes.cpu().numpy(), want["pos_bboxes"])
        np.testing.assert_array_equal(res.pos_gt_bboxes.cpu().numpy(), want["pos_gt_bboxes"])
        assert (res.pos_gt_labels.cpu().numpy() == want["pos_gt_labels"]).all()
        samples.append(want), results.append(res)
        # the fixed-size form of the same draw (what the train step runs): the same rows in the same order + masks
        ms = head.sampler.sample_masked(head.assigner.assign(pr, obb, None, lab0), pr, obb, lab0)
        npos, nneg = len(want["pos_inds"]), len(want["neg_inds"])
        assert int(ms.n_pos) == npos and int(ms.n_neg) == nneg and ms.inds.numel() == 512
        assert (ms.inds[:npos].cpu().numpy() == want["pos_inds"]).all()
        assert (ms.inds[npos:npos + nneg].cpu().numpy() == want["neg_inds"]).all()
        assert bool(ms.is_pos[:npos].all()) and not bool(ms.is_pos[npos:].any()) and bool(ms.valid[:npos + nneg].all())
        np.testing.assert_array_equal(ms.bboxes[:npos].cpu().numpy(), want["pos_bboxes"])
        np.testing.assert_array_equal(ms.pos_gt_bboxes[:npos].cpu().numpy(), want["pos_gt_bboxes"])
    assert len(samples[1]["pos_inds"]) == 128                          # the positive cap was hit: the choice mattered
    labels, lw, bt, _, bw = head.get_bboxes_targets(results)
    wl, wlw, wbt, wbw = H.np_oriented_head_targets(samples, cfg, 10)
    assert (labels.cpu().numpy() == wl).all() and (lw.cpu().numpy() == wlw).all() and (bw.cpu().numpy() == wbw).all()
    np.testing.assert_allclose(bt.cpu().numpy(), wbt, rtol=1e-4, atol=2e-4)
    rois = head.arb2roi([r.bboxes for r in results], bbox_type='obb')
    wr = H.np_oriented_head_rois(samples)
    np.testing.assert_array_equal(rois.cpu().numpy(), wr)
    n = len(wr)
    cls = rng.normal(0, 1.5, (n, 11)).astype(np.float32)
    reg = rng.normal(0, 0.5, (n, 5)).astype(np.float32)
    got = head.loss(torch.from_numpy(cls).to(cuda), torch.from_numpy(reg).to(cuda), rois, labels, lw, bt, None, bw)
    want = H.np_oriented_head_loss(cls, reg, wl, wlw, wbt, wbw, cfg, 10)
    for k in ("loss_cls", "orcnn_bbox_loss"):
        g, w = float(got[k]), want[k]
        assert w > 0 and abs(g - w) <= 2e-4 * abs(w) + 1e-6, (k, g, w)
    # test-time branch
    head.eval()
    r1 = wr[wr[:, 0] == 1]
    r1[:, 0] = 0
    sl = slice(len(wr) - len(r1), len(wr))
    det, dl = head.get_bboxes(torch.from_numpy(r1).to(cuda), torch.from_numpy(cls[sl]).to(cuda),
                              torch.from_numpy(reg[sl]).to(cuda), (size, size), 2.0, rescale=True)
    wd, wdl = H.np_oriented_head_get_bboxes(r1, cls[sl], reg[sl], 2.0, cfg, 10)
    assert det.shape == wd.shape and det.shape[0] > 500 and (dl.cpu().numpy() == wdl).all()
    np.testing.assert_allclose(det[:, 8].cpu().numpy(), wd[:, 8], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(det[:, :8].cpu().numpy(), wd[:, :8], rtol=1e-4, atol=5e-3)
