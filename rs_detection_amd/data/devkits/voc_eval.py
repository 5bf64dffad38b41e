"""DOTA Task-1 mAP (/root/reference/python/jdet/data/devkits/voc_eval.py:39-71, :236-336 and the per-class driver
/root/reference/python/jdet/data/dota.py:83-143).

``voc_eval_dota`` keeps the reference's greedy matching (detections in descending confidence, horizontal-hull gate
with the +1 pixel convention, polygon ``ovmax > ovthresh``, difficult / already-detected handling); the polygon
overlaps -- the only heavy part -- come from ONE GPU launch per class (all detection x ground-truth pairs of the
same image that pass the gate) instead of a Python loop over shapely calls."""
import numpy as np


def voc_ap(rec, prec, use_07_metric=False):
    """:39-71."""
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


def _hull(p):
    return p[:, 0::2].min(1), p[:, 1::2].min(1), p[:, 0::2].max(1), p[:, 1::2].max(1)


def _gated_overlaps(det_polys, det_img, gts, iou_matrix):
    """For every detection d: (indices of the gts of its image that pass the hull gate, their polygon IoUs)."""
    out = [None] * len(det_polys)
    pair_d, pair_box, slices = [], [], []
    for d, (bb, img) in enumerate(zip(det_polys, det_img)):
        R = gts.get(int(img))
        if R is None or R["box"].size == 0:
            out[d] = (np.zeros(0, np.int64), np.zeros(0))
            continue
        BBGT = R["box"].astype(float)
        gx1, gy1, gx2, gy2 = _hull(BBGT)
        bx1, by1, bx2, by2 = bb[0::2].min(), bb[1::2].min(), bb[0::2].max(), bb[1::2].max()
        iw = np.maximum(np.minimum(gx2, bx2) - np.maximum(gx1, bx1) + 1., 0.)
        ih = np.maximum(np.minimum(gy2, by2) - np.maximum(gy1, by1) + 1., 0.)
        inters = iw * ih
        uni = (bx2 - bx1 + 1.) * (by2 - by1 + 1.) + (gx2 - gx1 + 1.) * (gy2 - gy1 + 1.) - inters
        idx = np.where(inters / uni > 0)[0]          # :283-285
        slices.append((d, idx, len(pair_d)))
        pair_d.extend([d] * len(idx))
        pair_box.append(BBGT[idx])
    if pair_d:
        boxes = np.concatenate(pair_box)
        dets = det_polys[np.array(pair_d)]
        ious = iou_matrix(boxes, dets)               # iou_func(BBGT_keep[index], bb), :289-291
    for d, idx, start in slices:
        out[d] = (idx, ious[start:start + len(idx)] if len(idx) else np.zeros(0))
    return out


def _pairwise_gpu(device):
    """Row-wise polygon IoU of two equally long lists through the dense kernel, in chunks."""
    import torch
    from rs_detection_amd.ops import poly_iou_matrix

    def run(a, b, chunk=2048):
        res = np.empty(len(a))
        for s in range(0, len(a), chunk):
            m = poly_iou_matrix(a[s:s + chunk], b[s:s + chunk], device=device)
            res[s:s + chunk] = torch.diagonal(m).cpu().numpy()
        return res
    return run


def voc_eval_dota(dets, gts, iou_func=None, ovthresh=0.5, use_07_metric=False, device="cuda", pairwise=None):
    """:236-336.  dets (nd, 10) = image index, 8 polygon coordinates, confidence; gts {image index: dict(box (k,8),
    det [bool]*k, difficult bool (k,))}.  ``pairwise(a (m,8), b (m,8)) -> (m,)`` overrides the GPU polygon IoU
    (tests pass the CPU oracle); ``iou_func`` is accepted for signature parity and, when given, is called per pair
    like the reference does."""
    dets = np.array(np.asarray(dets).tolist(), dtype=np.float64).reshape(-1, 10)
    npos = sum(int(np.sum(~gts[k]["difficult"])) for k in gts)
    nd = len(dets)
    if nd == 0 or npos == 0:
        return 0., 0., 0.
    confidence = dets[:, -1]
    sorted_ind = np.argsort(-confidence)
    dets = dets[sorted_ind, :-1]
    if iou_func is not None:
        pairwise = lambda a, b: np.array([iou_func(x, y) for x, y in zip(a, b)])  # noqa: E731
    elif pairwise is None:
        pairwise = _pairwise_gpu(device)
    gated = _gated_overlaps(dets[:, 1:], dets[:, 0], gts, pairwise)
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d in range(nd):
        R = gts.get(int(dets[d, 0]))
        ovmax, jmax = -np.inf, -1
        idx, ov = gated[d]
        if len(idx) > 0:
            ovmax = np.max(ov)
            jmax = idx[int(np.argmax(ov))]
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


def evaluate_dota(results, classes, device="cuda", pairwise=None):
    """dota.py:83-143: results = [((polys (n,8), scores (n,), labels (n,) 0-based), target dict with ``polys``,
    ``labels`` (1-based), ``polys_ignore``, ``scale_factor``)] -> {"eval/<i>_<class>_AP": ap, "eval/0_meanAP": mAP}."""
    dets, gts, difficult_polys = [], [], {}
    for img_idx, (result, target) in enumerate(results):
        det_polys, det_scores, det_labels = (np.asarray(r) for r in result)
        det_labels = det_labels + 1
        if det_polys.size > 0:
            idx1 = np.ones((det_labels.shape[0], 1)) * img_idx
            dets.append(np.concatenate([idx1, det_polys.reshape(-1, 8), det_scores.reshape(-1, 1),
                                        det_labels.reshape(-1, 1)], axis=1))
        sf = target["scale_factor"]
        gt_polys = np.asarray(target["polys"], dtype=np.float64) / sf
        if gt_polys.size > 0:
            gt_labels = np.asarray(target["labels"]).reshape(-1, 1)
            gts.append(np.concatenate([np.ones((gt_labels.shape[0], 1)) * img_idx, gt_polys.reshape(-1, 8), gt_labels], axis=1))
        difficult_polys[img_idx] = np.asarray(target.get("polys_ignore", np.zeros((0, 8))), dtype=np.float64) / sf
    aps = {}
    if len(dets) == 0 or len(gts) == 0:
        for i, c in enumerate(classes):
            aps["eval/%d_%s_AP" % (i + 1, c)] = 0
        aps["eval/0_meanAP"] = 0
        return aps
    dets, gts = np.concatenate(dets), np.concatenate(gts)
    for i, c in enumerate(classes):
        c_dets = dets[dets[:, -1] == (i + 1)][:, :-1]
        c_gts = gts[gts[:, -1] == (i + 1)][:, :-1]
        classname_gts = {}
        for idx in np.unique(gts[:, 0]):
            g = c_gts[c_gts[:, 0] == idx, :][:, 1:]
            dg = difficult_polys[int(idx)].copy().reshape(-1, 8)
            difficulty = np.zeros(g.shape[0] + dg.shape[0], dtype=bool)
            difficulty[g.shape[0]:] = True
            g = np.concatenate([g, dg])
            classname_gts[int(idx)] = {"box": g.copy(), "det": [False] * len(g), "difficult": difficulty}
        _, _, ap = voc_eval_dota(c_dets, classname_gts, device=device, pairwise=pairwise)
        aps["eval/%d_%s_AP" % (i + 1, c)] = ap
    aps["eval/0_meanAP"] = sum(aps.values()) / len(aps)
    return aps
