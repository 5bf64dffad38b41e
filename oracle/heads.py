"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by rs_detection_amd/).

NumPy restatements of the head-level host logic of the hot path -- the glue between the kernels that the per-op
oracles do not cover.  Every function follows the reference lines it cites (under /root/reference/python/jdet/);
arithmetic in float32 in the reference's operation order.  Jittor itself is not importable here, so these
are line-by-line transcriptions, NOT reference outputs: "parity unpinned" for the argsort / argmax / topk tie rules
(first index, stable descending sort -- the torch semantics the build adopts), pinned for everything else by the
per-op oracles they call (IoU / NMS: the reference's own CPU source through oracle/_ref).

  np_s2anet_head_loss          models/roi_heads/s2anet_head.py:322-508 + boxes/anchor_target.py:18-180
  np_s2anet_get_bboxes_single  models/roi_heads/s2anet_head.py:543-601 + ops/nms_rotated.py:540-596
  np_obb2hbb / np_obb2poly / np_rectpoly2obb / np_regular_theta / np_regular_obb   ops/bbox_transforms.py:501-640
  np_midpoint_offset_encode / _decode        models/boxes/coder.py:327-433
  np_oriented_delta_encode / _decode         models/boxes/coder.py:444-513
  np_oriented_rpn_get_bboxes_single          models/roi_heads/oriented_rpn_head.py:156-227
"""
import numpy as np

import oracle

F = np.float32
PI = np.pi


# ---- ops/bbox_transforms.py ------------------------------------------------------------------------------------
def np_regular_theta(theta, mode='180', start=-PI / 2):
    """:501-507 (Python-style float mod, like Jittor's `%` on floats as adopted: SURVEY 8c)."""
    cycle = 2 * PI if mode == '360' else PI
    theta = (theta - F(start)).astype(F)
    theta = np.mod(theta, F(cycle)).astype(F)
    return (theta + F(start)).astype(F)


def np_regular_obb(obb):
    """:509-519: swap (w, h) so that w >= h, turn theta by pi/2 with it, wrap into [-pi/2, pi/2)."""
    x, y, w, h, t = [obb[..., k].astype(F) for k in range(5)]
    m = (w > h).astype(F)
    w_r = w * m + h * (1 - m)
    h_r = h * m + w * (1 - m)
    t_r = t * m + (t + F(PI / 2)) * (1 - m)
    return np.stack([x, y, w_r, h_r, np_regular_theta(t_r.astype(F))], -1).astype(F)


def np_obb2poly(obb):
    """:612-622."""
    c, w, h, t = obb[..., :2].astype(F), obb[..., 2:3].astype(F), obb[..., 3:4].astype(F), obb[..., 4:5].astype(F)
    cos, sin = np.cos(t).astype(F), np.sin(t).astype(F)
    v1 = np.concatenate([w / 2 * cos, -w / 2 * sin], -1)
    v2 = np.concatenate([-h / 2 * sin, -h / 2 * cos], -1)
    return np.concatenate([c + v1 + v2, c + v1 - v2, c - v1 - v2, c - v1 + v2], -1).astype(F)


def np_obb2hbb(obb):
    """:625-631."""
    c, w, h, t = obb[..., :2].astype(F), obb[..., 2:3].astype(F), obb[..., 3:4].astype(F), obb[..., 4:5].astype(F)
    cos, sin = np.cos(t).astype(F), np.sin(t).astype(F)
    bias = np.concatenate([np.abs(w / 2 * cos) + np.abs(h / 2 * sin), np.abs(w / 2 * sin) + np.abs(h / 2 * cos)], -1)
    return np.concatenate([c - bias, c + bias], -1).astype(F)


def np_rectpoly2obb(polys):
    """:577-599."""
    polys = polys.astype(F)
    theta = np.arctan2(-(polys[..., 3] - polys[..., 1]), polys[..., 2] - polys[..., 0]).astype(F)
    cos, sin = np.cos(theta).astype(F), np.sin(theta).astype(F)
    mat = np.stack([cos, -sin, sin, cos], -1).reshape(*theta.shape, 2, 2)
    x = polys[..., 0::2].mean(-1, dtype=F)
    y = polys[..., 1::2].mean(-1, dtype=F)
    center = np.stack([x, y], -1)[..., None, :]
    cp = polys.reshape(*polys.shape[:-1], 4, 2) - center
    rot = np.matmul(cp, np.swapaxes(mat, -1, -2)).astype(F)
    w = rot[..., :, 0].max(-1) - rot[..., :, 0].min(-1)
    h = rot[..., :, 1].max(-1) - rot[..., :, 1].min(-1)
    return np_regular_obb(np.stack([x, y, w, h, theta], -1).astype(F))


# ---- models/boxes/coder.py ---------------------------------------------------------------------------------------
def np_midpoint_offset_encode(bboxes, gt, means, stds):
    """MidpointOffsetCoder.encode :327-367: hbb proposal (x1,y1,x2,y2) + obb gt -> (dx,dy,dw,dh,da,db)."""
    p, gt = bboxes.astype(F), gt.astype(F)
    px, py = (p[..., 0] + p[..., 2]) * F(0.5), (p[..., 1] + p[..., 3]) * F(0.5)
    pw, ph = p[..., 2] - p[..., 0], p[..., 3] - p[..., 1]
    hbb, poly = np_obb2hbb(gt), np_obb2poly(gt)
    gx, gy = (hbb[..., 0] + hbb[..., 2]) * F(0.5), (hbb[..., 1] + hbb[..., 3]) * F(0.5)
    gw, gh = hbb[..., 2] - hbb[..., 0], hbb[..., 3] - hbb[..., 1]
    x_coor, y_coor = poly[:, 0::2], poly[:, 1::2]
    y_min = y_coor.min(1, keepdims=True)          # `_, y_min = y_coor.argmin(...)`: Jittor returns (index, VALUE)
    x_max = x_coor.max(1, keepdims=True)
    _x = x_coor.copy()
    _x[np.abs(y_coor - y_min) > 0.1] = -1000
    ga = _x.max(1)
    _y = y_coor.copy()
    _y[np.abs(x_coor - x_max) > 0.1] = -1000
    gb = _y.max(1)
    d = np.stack([(gx - px) / pw, (gy - py) / ph, np.log(gw / pw), np.log(gh / ph), (ga - gx) / gw, (gb - gy) / gh],
                 -1).astype(F)
    return ((d - np.asarray(means, F)[None]) / np.asarray(stds, F)[None]).astype(F)


def np_midpoint_offset_decode(bboxes, pred, means, stds, wh_ratio_clip=16 / 1000):
    """MidpointOffsetCoder.decode :369-433 -> obb (n, 5*k)."""
    bboxes, pred = bboxes.astype(F), pred.astype(F)
    rep = pred.shape[1] // 6
    d = pred * np.tile(np.asarray(stds, F), rep)[None] + np.tile(np.asarray(means, F), rep)[None]
    dx, dy, dw, dh, da, db = (d[:, k::6] for k in range(6))
    mr = F(np.abs(np.log(wh_ratio_clip)))
    dw, dh = np.clip(dw, -mr, mr), np.clip(dh, -mr, mr)
    px, py = ((bboxes[:, 0] + bboxes[:, 2]) * F(0.5))[:, None], ((bboxes[:, 1] + bboxes[:, 3]) * F(0.5))[:, None]
    pw, ph = (bboxes[:, 2] - bboxes[:, 0])[:, None], (bboxes[:, 3] - bboxes[:, 1])[:, None]
    gw, gh = pw * np.exp(dw), ph * np.exp(dh)
    gx, gy = px + pw * dx, py + ph * dy
    x1, y1, x2, y2 = gx - gw * F(0.5), gy - gh * F(0.5), gx + gw * F(0.5), gy + gh * F(0.5)
    da, db = np.clip(da, -0.5, 0.5), np.clip(db, -0.5, 0.5)
    ga, _ga, gb, _gb = gx + da * gw, gx - da * gw, gy + db * gh, gy - db * gh
    polys = np.stack([ga, y1, x2, gb, _ga, y2, x1, _gb], -1).astype(F)
    center = np.stack([gx, gy, gx, gy, gx, gy, gx, gy], -1).astype(F)
    cp = polys - center
    diag = np.sqrt(cp[..., 0::2] ** 2 + cp[..., 1::2] ** 2).astype(F)
    scale = diag.max(-1, keepdims=True) / diag     # `_, max_diag_len = diag_len.argmax(...)`: the VALUE
    cp = cp * np.repeat(scale, 2, -1)
    return np_rectpoly2obb((cp + center).astype(F)).reshape(pred.shape[0], -1).astype(F)


def np_oriented_delta_encode(bboxes, gt, means, stds):
    """OrientedDeltaXYWHTCoder.encode :452-482."""
    px, py, pw, ph, pt = [bboxes[..., k].astype(F) for k in range(5)]
    gx, gy, gw, gh, gt_ = [gt[..., k].astype(F) for k in range(5)]
    d1 = np_regular_theta((gt_ - pt).astype(F))
    d2 = np_regular_theta((gt_ - pt + F(PI / 2)).astype(F))
    m = (np.abs(d1) < np.abs(d2)).astype(F)
    gw_r = gw * m + gh * (1 - m)
    gh_r = gh * m + gw * (1 - m)
    dt = d1 * m + d2 * (1 - m)
    c, s = np.cos(-pt).astype(F), np.sin(-pt).astype(F)
    dx = (c * (gx - px) + s * (gy - py)) / pw
    dy = (-s * (gx - px) + c * (gy - py)) / ph
    d = np.stack([dx, dy, np.log(gw_r / pw), np.log(gh_r / ph), dt], -1).astype(F)
    return ((d - np.asarray(means, F)[None]) / np.asarray(stds, F)[None]).astype(F)


def np_oriented_delta_decode(bboxes, pred, means, stds, wh_ratio_clip=16 / 1000):
    """OrientedDeltaXYWHTCoder.decode :484-513 -> (n, 5*k)."""
    bboxes, pred = bboxes.astype(F), pred.astype(F)
    rep = pred.shape[1] // 5
    d = pred * np.tile(np.asarray(stds, F), rep)[None] + np.tile(np.asarray(means, F), rep)[None]
    dx, dy, dw, dh, dt = (d[:, k::5] for k in range(5))
    mr = F(np.abs(np.log(wh_ratio_clip)))
    dw, dh = np.clip(dw, -mr, mr), np.clip(dh, -mr, mr)
    px, py, pw, ph, pt = [np.broadcast_to(bboxes[:, k:k + 1], dx.shape) for k in range(5)]
    c, s = np.cos(-pt).astype(F), np.sin(-pt).astype(F)
    gx = dx * pw * c - dy * ph * s + px
    gy = dx * pw * s + dy * ph * c + py
    gw, gh = pw * np.exp(dw), ph * np.exp(dh)
    gt = np_regular_theta((dt + pt).astype(F))
    out = np_regular_obb(np.stack([gx, gy, gw, gh, gt], -1).astype(F))
    return out.reshape(pred.shape[0], -1).astype(F)


# ---- S2ANetHead.loss ---------------------------------------------------------------------------------------------
def _np_anchor_target_single(anchors, gts, labels, cfg):
    """boxes/anchor_target.py:105-180 for one image (sampling=False -> PseudoSampler, allowed_border=-1,
    pos_weight <= 0): -> labels, label_weights, bbox_targets, bbox_weights, #pos, #neg."""
    oc = oracle.c()
    A = anchors.shape[0]
    lab, lw = np.zeros(A, np.int32), np.zeros(A, F)
    bt, bw = np.zeros((A, 5), F), np.zeros((A, 5), F)
    ov = oc.box_iou_rotated(gts, anchors, 0)                                       # assigner.py:94
    gi, _, _ = oc.assign_wrt_overlaps(ov, cfg["pos_iou_thr"], cfg["neg_iou_thr"], cfg["min_pos_iou"], True, True,
                                      labels, 0)                                   # assigner.py:111-170
    pos, neg = np.nonzero(gi > 0)[0], np.nonzero(gi == 0)[0]                       # sampler.py:114-130
    if len(pos):
        bt[pos] = oracle.np_bbox2delta_rotated(anchors[pos], gts[gi[pos] - 1], cfg["means"], cfg["stds"])
        bw[pos] = 1.0
        lab[pos] = labels[gi[pos] - 1]
        lw[pos] = 1.0
    lw[neg] = 1.0
    return lab, lw, bt, bw, len(pos), len(neg)


def _np_level_losses(cls_scores, bbox_preds, labels, lw, bt, bw, num_level_anchors, avg, C, focal, smooth):
    """multi_apply(loss_*_single) :430-508: per pyramid level, all images of the batch flattened together."""
    l_cls, l_box, s = [], [], 0
    for lvl, n in enumerate(num_level_anchors):
        cs = np.transpose(cls_scores[lvl], (0, 2, 3, 1)).reshape(-1, C)            # :444-445
        bp = np.transpose(bbox_preds[lvl], (0, 2, 3, 1)).reshape(-1, 5)            # :451
        sl = slice(s, s + n)
        l_cls.append(oracle.np_sigmoid_focal_loss(cs, labels[:, sl].reshape(-1), lw[:, sl].reshape(-1),
                                                  focal["gamma"], focal["alpha"], avg) * focal.get("loss_weight", 1.0))
        l_box.append(oracle.np_smooth_l1_loss(bp, bt[:, sl].reshape(-1, 5), bw[:, sl].reshape(-1, 5), smooth["beta"],
                                              avg) * smooth.get("loss_weight", 1.0))
        s += n
    return l_cls, l_box


def np_s2anet_head_loss(fam_cls_scores, fam_bbox_preds, refine_anchors, odm_cls_scores, odm_bbox_preds, gt_bboxes,
                        gt_labels, strides, fam_cfg, odm_cfg, focal, smooth, num_classes=16):
    """S2ANetHead.loss :322-428 -> dict of four lists of per-level scalars.
    *_scores / *_preds: lists (levels) of (B, C|5, H, W) arrays; refine_anchors: list of (B, H, W, 5)."""
    C = num_classes - 1                                                            # cls_out_channels, :104-105
    B = fam_bbox_preds[0].shape[0]
    sizes = [p.shape[-2:] for p in odm_cls_scores]
    nla = [h * w for h, w in sizes]
    init = np.concatenate([oracle.np_s2anet_grid_anchors(sz, s) for sz, s in zip(sizes, strides)])   # :254-289
    refined = np.concatenate([r.reshape(B, -1, 5) for r in refine_anchors], 1)     # :291-320
    out = {}
    for tag, cfg, anchors_of, cs, bp in (("fam", fam_cfg, lambda b: init, fam_cls_scores, fam_bbox_preds),
                                         ("odm", odm_cfg, lambda b: refined[b], odm_cls_scores, odm_bbox_preds)):
        per = [_np_anchor_target_single(np.ascontiguousarray(anchors_of(b)), gt_bboxes[b], gt_labels[b], cfg)
               for b in range(B)]
        labels, lw = np.stack([p[0] for p in per]), np.stack([p[1] for p in per])
        bt, bw = np.stack([p[2] for p in per]), np.stack([p[3] for p in per])
        num_total_pos = sum(max(p[4], 1) for p in per)                             # anchor_target.py:79 (q11)
        l_cls, l_box = _np_level_losses(cs, bp, labels, lw, bt, bw, nla, float(num_total_pos), C, focal, smooth)
        out["loss_%s_cls" % tag], out["loss_%s_bbox" % tag] = l_cls, l_box
    return out


# ---- S2ANetHead.get_bboxes_single --------------------------------------------------------------------------------
def np_multiclass_nms_rotated(multi_bboxes, multi_scores, score_thr, iou_thr, max_num=-1):
    """ops/nms_rotated.py:540-596: class-wise candidates -> one class-aware NMS (`>=` of the CPU path, q1) -> kept
    rows in ascending index order (q8) re-sorted by score, `max_num` best."""
    num_classes = multi_scores.shape[1]
    if multi_bboxes.shape[1] > 5:
        bboxes = multi_bboxes.reshape(multi_scores.shape[0], -1, 5)[:, 1:]
    else:
        bboxes = np.broadcast_to(multi_bboxes[:, None], (multi_scores.shape[0], num_classes - 1, 5))
    scores = multi_scores[:, 1:]
    valid = scores > score_thr
    bboxes, scores = bboxes[valid], scores[valid]
    labels = np.nonzero(valid)[1]
    if bboxes.size == 0:
        return np.zeros((0, 6), F), np.zeros((0,), np.int64)
    dets6 = np.concatenate([bboxes, labels[:, None].astype(F)], 1).astype(F)
    order = np.argsort(-scores, kind="stable").astype(np.int32)
    keep = np.nonzero(oracle.c().nms_rotated(np.ascontiguousarray(dets6), order, iou_thr))[0]     # :527-538
    dets = np.concatenate([bboxes[keep], scores[keep, None]], 1).astype(F)
    labels = labels[keep]
    o = np.argsort(-dets[:, 5], kind="stable")                                     # :588-590
    if max_num > 0:
        o = o[:max_num]
    return dets[o], labels[o]


def np_s2anet_get_bboxes_single(cls_score_list, bbox_pred_list, mlvl_anchors, scale_factor, cfg, means, stds,
                                num_classes=16, rescale=True):
    """:543-601 for one image: lists (levels) of (C,H,W) / (5,H,W) / (H*W,5) -> polys (n,8), scores, labels."""
    C = num_classes - 1
    boxes, scores_all = [], []
    for cs, bp, an in zip(cls_score_list, bbox_pred_list, mlvl_anchors):
        s = (1.0 / (1.0 + np.exp(-np.transpose(cs, (1, 2, 0)).reshape(-1, C).astype(np.float64)))).astype(F)
        bp = np.transpose(bp, (1, 2, 0)).reshape(-1, 5)
        nms_pre = cfg.get("nms_pre", -1)
        if nms_pre > 0 and s.shape[0] > nms_pre:
            top = np.argsort(-s.max(1), kind="stable")[:nms_pre]                   # topk: descending, first index wins
            an, bp, s = an[top], bp[top], s[top]
        boxes.append(oracle.np_delta2bbox_rotated(an, bp, means, stds))            # :579-580
        scores_all.append(s)
    boxes, scores_all = np.concatenate(boxes), np.concatenate(scores_all)
    if rescale:
        boxes[:, :4] /= F(scale_factor)
    scores_all = np.concatenate([np.zeros((scores_all.shape[0], 1), F), scores_all], 1)   # dummy background, :588-591
    dets, labels = np_multiclass_nms_rotated(boxes, scores_all, cfg["score_thr"], cfg["nms"]["iou_thr"],
                                             cfg["max_per_img"])
    return oracle.np_rotated_box_to_poly(dets[:, :5]), dets[:, 5], labels


# ---- OrientedRPNHead._get_bboxes_single --------------------------------------------------------------------------
def np_hbb_nms(dets, thr):
    """jt.nms semantics as adopted (third-party, unpinned): greedy, score-descending (stable), IoU with the
    legacy +1 pixel convention, suppress on IoU > thr.  Returns kept indices in score order."""
    order = np.argsort(-dets[:, 4], kind="stable")
    area = (dets[:, 2] - dets[:, 0] + 1) * (dets[:, 3] - dets[:, 1] + 1)
    dead = np.zeros(len(dets), bool)
    keep = []
    for a, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(i)
        rest = order[a + 1:]
        w = np.maximum(0.0, np.minimum(dets[i, 2], dets[rest, 2]) - np.maximum(dets[i, 0], dets[rest, 0]) + 1)
        h = np.maximum(0.0, np.minimum(dets[i, 3], dets[rest, 3]) - np.maximum(dets[i, 1], dets[rest, 1]) + 1)
        inter = (w * h).astype(F)
        dead[rest[inter / (area[i] + area[rest] - inter) > thr]] = True
    return np.array(keep, np.int64)


def np_oriented_rpn_get_bboxes_single(cls_scores, bbox_preds, mlvl_anchors, means, stds, nms_pre, nms_post, nms_thresh,
                                      min_bbox_size=0, reg_dim=6):
    """:156-227 (sigmoid scores): per level top-`nms_pre`, MidpointOffset decode, per-level-offset hbb NMS,
    `nms_post` best -> (n, 6) = obb + score."""
    sc, bp, an, ids = [], [], [], []
    for idx, (cs, bpred, anchors) in enumerate(zip(cls_scores, bbox_preds, mlvl_anchors)):
        s = (1.0 / (1.0 + np.exp(-np.transpose(cs, (1, 2, 0)).reshape(-1).astype(np.float64)))).astype(F)
        b = np.transpose(bpred, (1, 2, 0)).reshape(-1, reg_dim)
        if nms_pre > 0 and s.shape[0] > nms_pre:
            top = np.argsort(-s, kind="stable")[:nms_pre]
            s, b, anchors = s[top], b[top], anchors[top]
        sc.append(s), bp.append(b), an.append(anchors), ids.append(np.full(s.shape[0], idx, np.int64))
    sc, bp, an, ids = np.concatenate(sc), np.concatenate(bp), np.concatenate(an), np.concatenate(ids)
    props = np_midpoint_offset_decode(an, bp, means, stds)
    if min_bbox_size >= 0:
        ok = (props[:, 2] > min_bbox_size) & (props[:, 3] > min_bbox_size)
        if not ok.all():
            props, sc, ids = props[ok], sc[ok], ids[ok]
    hp = np_obb2hbb(props)
    hp = hp + (ids.astype(F) * (hp.max() - hp.min() + 1))[:, None]                 # levels never suppress each other
    keep = np_hbb_nms(np.concatenate([hp, sc[:, None]], 1).astype(F), nms_thresh)
    return np.concatenate([props, sc[:, None]], 1)[keep][:nms_post].astype(F)


# ---- Oriented R-CNN: shared pieces -------------------------------------------------------------------------------
def np_bbox_overlaps_hbb(b1, b2, version=0, eps=1e-6):
    """models/boxes/iou_calculator.py:164-270, mode 'iou', not aligned: (m,4) x (n,4) -> (m,n) fp32."""
    b1, b2 = b1[:, :4].astype(F), b2[:, :4].astype(F)
    a1 = (b1[:, 2] - b1[:, 0] + F(version)) * (b1[:, 3] - b1[:, 1] + F(version))
    a2 = (b2[:, 2] - b2[:, 0] + F(version)) * (b2[:, 3] - b2[:, 1] + F(version))
    lt = np.maximum(b1[:, None, :2], b2[None, :, :2])
    rb = np.minimum(b1[:, None, 2:], b2[None, :, 2:])
    wh = np.clip(rb - lt + F(version), 0, None).astype(F)
    ov = wh[..., 0] * wh[..., 1]
    union = np.maximum(a1[:, None] + a2[None, :] - ov, F(eps))
    return (ov / union).astype(F)


def np_hbb2obb(h):
    """ops/bbox_transforms.py:640-652."""
    x, y = (h[..., 0] + h[..., 2]) * F(0.5), (h[..., 1] + h[..., 3]) * F(0.5)
    w, hh = h[..., 2] - h[..., 0], h[..., 3] - h[..., 1]
    z = np.zeros_like(x)
    o1 = np.stack([x, y, w, hh, z], -1)
    o2 = np.stack([x, y, hh, w, z - F(PI / 2)], -1)
    flag = (w >= hh)[..., None].astype(F)
    return (flag * o1 + (1 - flag) * o2).astype(F)


def np_anchor_generator_grid(strides, ratios, scales, featmap_sizes):
    """AnchorGenerator (models/boxes/anchor_generator.py:94-407): base anchors (:232-272: scale_major, centre offset 0,
    base size = stride) + grid (:386-407: x fastest, the A base anchors of a cell adjacent) -> list of (H*W*A, 4)."""
    out = []
    r, s = np.asarray(ratios, F), np.asarray(scales, F)
    for st, (fh, fw) in zip(strides, featmap_sizes):
        hr = np.sqrt(r).astype(F)
        wr = (F(1) / hr).astype(F)
        ws = (F(st) * wr[:, None] * s[None, :]).reshape(-1)
        hs = (F(st) * hr[:, None] * s[None, :]).reshape(-1)
        base = np.stack([-F(0.5) * ws, -F(0.5) * hs, F(0.5) * ws, F(0.5) * hs], -1).astype(F)
        sx, sy = np.arange(fw, dtype=F) * st, np.arange(fh, dtype=F) * st
        xx, yy = np.tile(sx, fh), np.repeat(sy, fw)
        shifts = np.stack([xx, yy, xx, yy], -1)
        out.append((base[None] + shifts[:, None]).reshape(-1, 4).astype(F))
    return out


def np_anchor_valid_flags(strides, featmap_sizes, pad_shape, num_base):
    """:425-470: cells whose origin lies inside the padded image, repeated for the A base anchors."""
    out = []
    for st, (fh, fw) in zip(strides, featmap_sizes):
        vh, vw = min(int(np.ceil(pad_shape[0] / st)), fh), min(int(np.ceil(pad_shape[1] / st)), fw)
        vx, vy = np.arange(fw) < vw, np.arange(fh) < vh
        out.append(np.repeat(np.tile(vx, fh) & np.repeat(vy, fw), num_base))
    return out


def np_anchor_inside_flags(flat_anchors, valid, img_shape, allowed_border=0):
    """models/boxes/anchor_target.py:184-195 (img_shape[:2] read as (h, w))."""
    h, w = img_shape[:2]
    if allowed_border < 0:
        return valid
    a = flat_anchors
    return valid & (a[:, 0] >= -allowed_border) & (a[:, 1] >= -allowed_border) & \
        (a[:, 2] < w + allowed_border) & (a[:, 3] < h + allowed_border)


def np_random_sample(gt_inds, num, pos_fraction, neg_pos_ub, choice):
    """BaseSampler.sample + RandomSampler._sample_pos/_neg (models/boxes/sampler.py:57-111,156-176) on the (already
    gt-extended) gt_inds.  ``choice(gallery, n)`` stands in for ``gallery[jt.randperm(len)[:n]]`` (:139-149): the test
    feeds the SAME fixed choice to both sides.  `.unique()` (:94,107) sorts."""
    pos = np.nonzero(gt_inds > 0)[0]
    n_pos = int(num * pos_fraction)
    if len(pos) > n_pos:
        pos = choice(pos, n_pos)
    pos = np.unique(pos)
    n_neg = num - len(pos)
    if neg_pos_ub >= 0:
        n_neg = min(n_neg, int(neg_pos_ub * max(1, len(pos))))
    neg = np.nonzero(gt_inds == 0)[0]
    if len(neg) > n_neg:
        neg = choice(neg, n_neg)
    return pos, np.unique(neg)


def np_bce_with_logits_sum(pred, target, weight):
    """jt.nn.binary_cross_entropy_with_logits(size_average=False) as used by losses/cross_entropy_loss.py:24-32."""
    x, t = pred.astype(np.float64), target.astype(np.float64)
    return float(((np.maximum(x, 0) - x * t + np.log1p(np.exp(-np.abs(x)))) * weight).sum())


def np_cross_entropy_rows(pred, target):
    """losses/cross_entropy_loss.py:57-66: logsumexp(pred - max) - (pred - max)[target] per row."""
    x = pred.astype(np.float64)
    x = x - x.max(1, keepdims=True)
    return np.log(np.exp(x).sum(1)) - x[np.arange(len(x)), target]


def np_smooth_l1_sum(pred, target, weight, beta):
    """losses/smooth_l1_loss.py:5-16 before the reduction."""
    d = np.abs(pred.astype(np.float64) - target.astype(np.float64))
    loss = np.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta) if beta != 0 else d
    return float((loss * weight).sum())


# ---- OrientedRPNHead.loss ----------------------------------------------------------------------------------------
def np_oriented_rpn_targets_single(flat_anchors, valid, gt_obb, img_size, cfg, choice):
    """_get_targets_single (roi_heads/oriented_rpn_head.py:274-366) for one image.  ``gt_obb``: the target's rboxes
    BEFORE the sign flip of :281-282 (done here).  -> labels, label_weights, bbox_targets, bbox_weights (all anchors),
    pos_inds, neg_inds (indices into the INSIDE subset, as the reference returns them)."""
    gt = gt_obb.astype(F).copy()
    gt[:, -1] *= -1
    inside = np_anchor_inside_flags(flat_anchors, valid, img_size, 0)              # :295
    anchors = flat_anchors[inside]
    tgt = np_obb2hbb(gt)                                                           # bbox2type(gt, 'hbb') :302
    ov = np_bbox_overlaps_hbb(tgt, anchors)                                        # assigner.py:94, BboxOverlaps2D
    a = cfg["assigner"]
    gi, _, _ = oracle.c().assign_wrt_overlaps(ov, a["pos_iou_thr"], a["neg_iou_thr"], a["min_pos_iou"],
                                              a.get("match_low_quality", True), True, None,
                                              a.get("assigned_labels_filled", -1))
    s = cfg["sampler"]
    assert not s.get("add_gt_as_proposals", False)
    pos, neg = np_random_sample(gi, s["num"], s["pos_fraction"], s.get("neg_pos_ub", -1), choice)   # :307
    n = anchors.shape[0]
    bt, bw = np.zeros((n, 6), F), np.zeros((n, 6), F)
    lab, lw = np.full(n, cfg.get("background_label", 0), np.int64), np.zeros(n, F)
    if len(pos):
        pos_gt = gt[gi[pos] - 1]                                                   # :310-314: the OBB gts
        c = cfg["bbox_coder"]
        bt[pos] = np_midpoint_offset_encode(anchors[pos], pos_gt, c["target_means"], c["target_stds"])   # :327-329
        bw[pos] = 1.0
        lab[pos] = 1                                                               # :334-336
        lw[pos] = 1.0 if cfg.get("pos_weight", -1) <= 0 else cfg["pos_weight"]
    lw[neg] = 1.0

    def unmap(d, fill=0):                                                          # :349-361
        out = np.full((flat_anchors.shape[0],) + d.shape[1:], fill, d.dtype)
        out[inside] = d
        return out
    return unmap(lab, cfg.get("background_label", 0)), unmap(lw), unmap(bt), unmap(bw), pos, neg


def np_oriented_rpn_loss(cls_scores, bbox_preds, targets, cfg, choice):
    """OrientedRPNHead.loss (:432-480) + get_targets (:368-398) + loss_single (:400-430) ->
    dict(loss_rpn_cls=[per level], loss_rpn_bbox=[per level]).  cls_scores / bbox_preds: lists (levels) of
    (B, A*1, H, W) / (B, A*6, H, W); targets: list of dict(rboxes, img_size, pad_shape)."""
    ag = cfg["anchor_generator"]
    sizes = [tuple(c.shape[-2:]) for c in cls_scores]
    mla = np_anchor_generator_grid(ag["strides"], ag["ratios"], ag["scales"], sizes)
    nA = len(ag["ratios"]) * len(ag["scales"])
    flat = np.concatenate(mla)
    per = []
    for t in targets:
        valid = np.concatenate(np_anchor_valid_flags(ag["strides"], sizes, t["pad_shape"], nA))
        per.append(np_oriented_rpn_targets_single(flat, valid, t["rboxes"], t["img_size"], cfg, choice))
    num_total = sum(max(len(p[4]), 1) for p in per) + sum(max(len(p[5]), 1) for p in per)      # :386-387, :466
    labels, lw = np.stack([p[0] for p in per]), np.stack([p[1] for p in per])
    bt, bw = np.stack([p[2] for p in per]), np.stack([p[3] for p in per])
    beta = cfg["loss_bbox"].get("beta", 1.0)
    l_cls, l_box, s = [], [], 0
    for lvl, a in enumerate(mla):
        n = a.shape[0]
        cs = np.transpose(cls_scores[lvl], (0, 2, 3, 1)).reshape(-1, 1)             # :424: cls_out_channels = 1
        bp = np.transpose(bbox_preds[lvl], (0, 2, 3, 1)).reshape(-1, 6)
        lb, w = labels[:, s:s + n].reshape(-1), lw[:, s:s + n].reshape(-1)
        # _expand_binary_labels (cross_entropy_loss.py:15-23): column label-1 set where label >= 1
        l_cls.append(cfg["loss_cls"].get("loss_weight", 1.0) *
                     np_bce_with_logits_sum(cs, (lb >= 1).astype(F)[:, None], w[:, None]) / num_total)
        l_box.append(cfg["loss_bbox"].get("loss_weight", 1.0) *
                     np_smooth_l1_sum(bp, bt[:, s:s + n].reshape(-1, 6), bw[:, s:s + n].reshape(-1, 6), beta) / num_total)
        s += n
    return dict(loss_rpn_cls=l_cls, loss_rpn_bbox=l_box), per


# ---- OrientedHead ------------------------------------------------------------------------------------------------
def np_oriented_head_sample(proposals, gt_obb, gt_labels_1based, cfg, choice):
    """OrientedHead.execute (roi_heads/oriented_head.py:539-588) for one image: sign flip of theta + 0-based labels
    (:551-552,:564), MaxIoUAssigner over rotated IoU v1 (K x P), RandomSamplerRotated with the gts prepended
    (sampler.py:204-232).  -> dict(pos_bboxes, neg_bboxes, pos_gt_bboxes, pos_gt_labels, pos_inds, neg_inds)."""
    gt = gt_obb.astype(F).copy()
    gt[:, -1] *= -1
    labels = gt_labels_1based.astype(np.int32) - 1
    props = proposals.astype(F)
    a = cfg["assigner"]
    ov = oracle.c().box_iou_rotated(gt, np.ascontiguousarray(props[:, :5]), 1)
    gi, _, lb = oracle.c().assign_wrt_overlaps(ov, a["pos_iou_thr"], a["neg_iou_thr"], a["min_pos_iou"],
                                               a.get("match_low_quality", True), True, labels,
                                               a.get("assigned_labels_filled", -1))
    s = cfg["sampler"]
    boxes = props[:, :5]
    if s.get("add_gt_as_proposals", True):                                         # sampler.py:214-218 + add_gt_
        boxes = np.concatenate([gt, boxes])
        gi = np.concatenate([np.arange(1, len(gt) + 1, dtype=gi.dtype), gi])
        lb = np.concatenate([labels, lb])
    pos, neg = np_random_sample(gi, s["num"], s["pos_fraction"], s.get("neg_pos_ub", -1), choice)
    return dict(pos_bboxes=boxes[pos], neg_bboxes=boxes[neg], pos_gt_bboxes=gt[gi[pos] - 1], pos_gt_labels=lb[pos],
                pos_inds=pos, neg_inds=neg)


def np_oriented_head_targets(samples, cfg, num_classes):
    """get_bboxes_targets / get_bboxes_target_single (:426-496), reg_decoded_bbox False, concat=True."""
    c = cfg["bbox_coder"]
    labs, lws, bts, bws = [], [], [], []
    for s in samples:
        npos, nneg = len(s["pos_bboxes"]), len(s["neg_bboxes"])
        n = npos + nneg
        lab, lw = np.full(n, num_classes, np.int64), np.zeros(n, F)
        bt, bw = np.zeros((n, 5), F), np.zeros((n, 5), F)
        if npos:
            lab[:npos] = s["pos_gt_labels"]
            lw[:npos] = 1.0 if cfg.get("pos_weight", -1) <= 0 else cfg["pos_weight"]
            bt[:npos] = np_oriented_delta_encode(s["pos_bboxes"], s["pos_gt_bboxes"], c["target_means"], c["target_stds"])
            bw[:npos] = 1
        if nneg:
            lw[-nneg:] = 1.0
        labs.append(lab), lws.append(lw), bts.append(bt), bws.append(bw)
    return np.concatenate(labs), np.concatenate(lws), np.concatenate(bts), np.concatenate(bws)


def np_oriented_head_rois(samples):
    """arb2roi (:259-277) over [pos_bboxes ; neg_bboxes] of each image: (n, 6) = (batch index, obb)."""
    return np.concatenate([np.concatenate([np.full((len(s["pos_bboxes"]) + len(s["neg_bboxes"]), 1), i, F),
                                           np.concatenate([s["pos_bboxes"], s["neg_bboxes"]])], 1)
                           for i, s in enumerate(samples)]).astype(F)


def np_oriented_head_loss(cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, cfg, num_classes):
    """OrientedHead.loss (:354-424), class-agnostic regression, CrossEntropyLoss 'mean' with
    avg_factor = #(label_weights > 0) (:358-365) and SmoothL1 over the positives / avg_factor = #samples (:412-417)."""
    avg = max(float((label_weights > 0).sum()), 1.0)
    lc = cfg["loss_cls"].get("loss_weight", 1.0) * float((np_cross_entropy_rows(cls_score, labels) * label_weights).sum()) / avg
    pos = (labels >= 0) & (labels < num_classes)
    if pos.any():
        lb = cfg["loss_bbox"].get("loss_weight", 1.0) * np_smooth_l1_sum(
            bbox_pred.reshape(len(bbox_pred), 5)[pos], bbox_targets[pos], bbox_weights[pos],
            cfg["loss_bbox"].get("beta", 1.0)) / bbox_targets.shape[0]
    else:
        lb = 0.0
    return dict(loss_cls=lc, orcnn_bbox_loss=lb)


def np_oriented_head_get_bboxes(rois, cls_score, bbox_pred, scale_factor, cfg, num_classes):
    """get_bboxes (:498-536) + get_results (:279-305): softmax, OrientedDeltaXYWHT decode (max_shape is accepted and
    unused by the coder, coder.py:484-513), rescale of (x, y, w, h), score threshold over the foreground columns ->
    (n, 9) polys + score in (roi, class) row-major order of the mask, labels (n,)."""
    x = cls_score.astype(np.float64)
    e = np.exp(x - x.max(1, keepdims=True))
    scores = (e / e.sum(1, keepdims=True)).astype(F)
    c = cfg["bbox_coder"]
    boxes = np_oriented_delta_decode(rois[:, 1:], bbox_pred, c["target_means"], c["target_stds"])
    sf = np.asarray([scale_factor] * 4 if isinstance(scale_factor, float) else scale_factor, F)
    boxes = boxes.reshape(len(boxes), -1, 5)
    boxes = np.concatenate([boxes[..., :4] / sf, boxes[..., 4:]], -1).reshape(len(boxes), -1)
    b = np.broadcast_to(boxes[:, None], (len(boxes), num_classes, 5))
    s = scores[:, :-1]
    valid = s > cfg["score_thresh"]
    return np.concatenate([np_obb2poly(b[valid]), s[valid][:, None]], 1).astype(F), np.nonzero(valid)[1]
