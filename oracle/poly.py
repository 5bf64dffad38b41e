"""CPU oracle for the evaluation-side polygon ops (TEST INFRASTRUCTURE ONLY; nothing under rs_detection_amd/ may
import this).  Restates, in float64 NumPy / plain Python:
  * iou_poly                 /root/reference/python/jdet/ops/nms_poly.py:247-252 -- the reference calls shapely
                             (third-party, pinned 1.8.2 in requirements.txt, absent here): its published semantics
                             (area of the set intersection of two simple polygons) are restated with
                             Sutherland-Hodgman clipping, valid when the clipper is convex.  PARITY UNPINNED.
  * py_cpu_nms_poly_fast     /root/reference/python/jdet/data/devkits/result_merge.py:66-126, line by line
                             (with a stable descending sort instead of ``argsort()[::-1]``).
  * voc_eval_dota            /root/reference/python/jdet/data/devkits/voc_eval.py:236-336, line by line."""
import numpy as np


def _area(p):
    x, y = p[:, 0], p[:, 1]
    return 0.5 * float(np.dot(x, np.roll(y, -1)) - np.dot(np.roll(x, -1), y))


def _ccw(q):
    p = np.asarray(q, np.float64).reshape(4, 2)
    return p[::-1].copy() if _area(p) < 0 else p.copy()


def _convex(p):
    for i in range(4):
        a, b, c = p[i], p[(i + 1) % 4], p[(i + 2) % 4]
        if (b[0] - a[0]) * (c[1] - b[1]) - (b[1] - a[1]) * (c[0] - b[0]) < 0:
            return False
    return True


def _clip_area(subject, clipper):
    cur = [tuple(v) for v in subject]
    for e in range(4):
        a, b = clipper[e], clipper[(e + 1) % 4]
        ex, ey = b[0] - a[0], b[1] - a[1]
        nxt = []
        for i in range(len(cur)):
            p, q = cur[i], cur[(i + 1) % len(cur)]
            sp = ex * (p[1] - a[1]) - ey * (p[0] - a[0])
            sq = ex * (q[1] - a[1]) - ey * (q[0] - a[0])
            if sp >= 0:
                nxt.append(p)
            if (sp >= 0) != (sq >= 0):
                t = sp / (sp - sq)
                nxt.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
        cur = nxt
        if not cur:
            return 0.0
    if len(cur) < 3:
        return 0.0
    return abs(_area(np.array(cur)))


def iou_poly(poly1, poly2):
    p1, p2 = _ccw(poly1), _ccw(poly2)
    inter = _clip_area(p1, p2) if _convex(p2) else _clip_area(p2, p1)
    return inter / max(abs(_area(p1)) + abs(_area(p2)) - inter, 0.01)


def py_cpu_nms_poly_fast(dets, thresh):
    dets = np.asarray(dets, np.float64)
    obbs = dets[:, 0:-1]
    x1, y1 = np.min(obbs[:, 0::2], axis=1), np.min(obbs[:, 1::2], axis=1)
    x2, y2 = np.max(obbs[:, 0::2], axis=1), np.max(obbs[:, 1::2], axis=1)
    scores = dets[:, 8]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    polys = [dets[i, :8].copy() for i in range(len(dets))]
    order = np.argsort(-scores, kind="stable")
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        xx1, yy1 = np.maximum(x1[i], x1[order[1:]]), np.maximum(y1[i], y1[order[1:]])
        xx2, yy2 = np.minimum(x2[i], x2[order[1:]]), np.minimum(y2[i], y2[order[1:]])
        w, h = np.maximum(0.0, xx2 - xx1), np.maximum(0.0, yy2 - yy1)
        hbb_inter = w * h
        hbb_ovr = hbb_inter / (areas[i] + areas[order[1:]] - hbb_inter)
        h_inds = np.where(hbb_ovr > 0)[0]
        tmp_order = order[h_inds + 1]
        for j in range(tmp_order.size):
            hbb_ovr[h_inds[j]] = iou_poly(polys[i], polys[tmp_order[j]])
        inds = np.where(hbb_ovr <= thresh)[0]
        order = order[inds + 1]
    return keep


def voc_ap(rec, prec, use_07_metric=False):
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap = ap + p / 11.
        return ap
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def voc_eval_dota(dets, gts, iou_func=iou_poly, ovthresh=0.5, use_07_metric=False):
    dets = np.array(np.asarray(dets).tolist(), dtype=np.float64).reshape(-1, 10)
    npos = sum([sum(~gts[k]["difficult"]) for k in gts])
    nd = len(dets)
    if nd == 0 or npos == 0:
        return 0., 0., 0.
    confidence = dets[:, -1]
    dets = dets[:, :-1]
    sorted_ind = np.argsort(-confidence)
    dets = dets[sorted_ind, :]
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d, det in enumerate(dets):
        bb = det[1:].astype(float)
        ovmax = -np.inf
        R = gts[int(det[0])]
        BBGT = R["box"].astype(float)
        if BBGT.size > 0:
            BBGT_xmin, BBGT_ymin = np.min(BBGT[:, 0::2], axis=1), np.min(BBGT[:, 1::2], axis=1)
            BBGT_xmax, BBGT_ymax = np.max(BBGT[:, 0::2], axis=1), np.max(BBGT[:, 1::2], axis=1)
            bb_xmin, bb_ymin, bb_xmax, bb_ymax = np.min(bb[0::2]), np.min(bb[1::2]), np.max(bb[0::2]), np.max(bb[1::2])
            iw = np.maximum(np.minimum(BBGT_xmax, bb_xmax) - np.maximum(BBGT_xmin, bb_xmin) + 1., 0.)
            ih = np.maximum(np.minimum(BBGT_ymax, bb_ymax) - np.maximum(BBGT_ymin, bb_ymin) + 1., 0.)
            inters = iw * ih
            uni = ((bb_xmax - bb_xmin + 1.) * (bb_ymax - bb_ymin + 1.) +
                   (BBGT_xmax - BBGT_xmin + 1.) * (BBGT_ymax - BBGT_ymin + 1.) - inters)
            overlaps = inters / uni
            BBGT_keep = BBGT[overlaps > 0, :]
            BBGT_keep_index = np.where(overlaps > 0)[0]
            if len(BBGT_keep) > 0:
                overlaps = [iou_func(BBGT_keep[index], bb) for index in range(len(BBGT_keep))]
                ovmax = np.max(overlaps)
                jmax = BBGT_keep_index[np.argmax(overlaps)]
        if ovmax > ovthresh:
            if not R['difficult'][jmax]:
                if not R['det'][jmax]:
                    tp[d] = 1.
                    R['det'][jmax] = 1
                else:
                    fp[d] = 1.
        else:
            fp[d] = 1.
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)
