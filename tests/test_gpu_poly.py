"""GPU, SURVEY 8f rank 1: polygon IoU kernel, polygon NMS (py_cpu_nms_poly_fast), mergebypoly on files and the DOTA mAP
driver -- each against the line-by-line CPU restatements in oracle/poly.py."""
import os

import numpy as np
import pytest
import torch

from oracle import poly as opoly
from test_devkits_cpu import _rbox_poly, _sq, _synthetic_eval_set

pytestmark = pytest.mark.gpu


def _random_quads(rng, n, span=600, convex_only=True):
    out = []
    for _ in range(n):
        if convex_only or rng.random() < 0.8:
            q = _rbox_poly(*rng.uniform(0, span, 2), rng.uniform(5, 150), rng.uniform(5, 80), rng.uniform(-np.pi, np.pi))
            q = q + rng.normal(0, 1.5, 8)                     # a general convex quadrilateral, not a rectangle
            if rng.random() < 0.5:
                q = q.reshape(4, 2)[::-1].reshape(-1)         # clockwise
        else:
            x, y, s = rng.uniform(0, span), rng.uniform(0, span), rng.uniform(20, 100)
            q = np.array([x, y, x + s, y + s / 2, x, y + s, x + 0.4 * s, y + s / 2])   # arrow head (concave)
        out.append(q)
    return np.stack(out)


def test_poly_iou_matrix_vs_oracle(cuda):
    from rs_detection_amd.ops import poly_iou_matrix, iou_poly
    rng = np.random.default_rng(0)
    a, b = _random_quads(rng, 70), _random_quads(rng, 90, convex_only=False)
    got = poly_iou_matrix(a, b, device=cuda).cpu().numpy()
    want = np.array([[opoly.iou_poly(x, y) for y in b] for x in a])
    assert np.abs(got - want).max() <= 1e-9
    assert ((got == 0) == (want == 0)).all()
    assert iou_poly(_sq(0, 0, 10), _sq(5, 0, 10), device=cuda) == pytest.approx(50 / 150, abs=1e-15)
    assert iou_poly(_sq(0, 0, 10), [5, -5, 15, 5, 5, 15, -5, 5], device=cuda) == pytest.approx(0.5, abs=1e-15)
    same = poly_iou_matrix(a, a, device=cuda).cpu().numpy()
    assert np.abs(np.diag(same) - 1).max() <= 1e-12 and np.abs(same - same.T).max() <= 1e-9   # self-IoU, symmetry
    assert poly_iou_matrix(np.zeros((0, 8)), a, device=cuda).shape == (0, 70)


@pytest.mark.parametrize("n,thr", [(1, 0.1), (64, 0.1), (65, 0.3), (700, 0.1), (2500, 0.3)])
def test_nms_poly_vs_py_cpu_nms_poly_fast(cuda, n, thr):
    from rs_detection_amd.ops import nms_poly
    rng = np.random.default_rng(n)
    centres = rng.uniform(0, 2000, (max(n // 8, 1), 2))
    c = centres[rng.integers(0, len(centres), n)] + rng.normal(0, 4, (n, 2))
    polys = np.stack([_rbox_poly(c[i, 0], c[i, 1], rng.uniform(30, 90), rng.uniform(10, 40), rng.uniform(-1.5, 1.5))
                      for i in range(n)])
    dets = np.concatenate([polys, rng.uniform(0.05, 1, (n, 1))], 1)
    want = opoly.py_cpu_nms_poly_fast(dets, thr)
    got = nms_poly(dets, thr, device=cuda).cpu().numpy()
    assert got.tolist() == [int(i) for i in want]          # same boxes in the same (descending score) order
    assert nms_poly(np.zeros((0, 9)), thr, device=cuda).numel() == 0


def test_mergebypoly_files_and_map_driver_on_gpu(cuda, tmp_path):
    from rs_detection_amd.data.devkits import mergebypoly, evaluate_dota
    rng = np.random.default_rng(9)
    src, dst = tmp_path / "before_nms", tmp_path / "after_nms"
    src.mkdir()
    lines = {}
    for cls in ("Ship", "Bridge"):
        rows = []
        for img in ("P1", "P2"):
            for _ in range(60):
                p = _rbox_poly(*rng.uniform(100, 1800, 2), rng.uniform(30, 80), rng.uniform(10, 30), rng.uniform(-1, 1))
                for dx, dy in ((0, 0), (824, 0)):          # every object reported by two overlapping tiles
                    rows.append(("%s__1__%d___%d" % (img, dx, dy), "%.4f" % rng.uniform(0.1, 1),
                                 ["%.4f" % v for v in (p - np.tile([dx, dy], 4) + rng.normal(0, 0.5, 8))]))
        lines[cls] = rows
        with open(src / (cls + ".txt"), "w") as f:
            for name, score, poly in rows:
                f.write(" ".join([name, score] + poly) + "\n")
    mergebypoly(str(src), str(dst), device=cuda)
    from rs_detection_amd.data.devkits import merge_detections
    for cls in lines:
        want = merge_detections(lines[cls], 0.1, lambda d, t: opoly.py_cpu_nms_poly_fast(d, t))
        got = {}
        for l in open(dst / (cls + ".txt")).read().strip().splitlines():
            t = l.split()
            got.setdefault(t[0], []).append([float(v) for v in t[2:]] + [float(t[1])])
        assert set(got) == set(want)
        for img in want:
            np.testing.assert_allclose(np.array(got[img]), np.array(want[img]), rtol=0, atol=1e-9)
            assert 55 <= len(got[img]) <= 70               # duplicates from the second tile are gone
    # DOTA mAP: GPU pairwise IoU == oracle pairwise IoU
    results = _synthetic_eval_set(np.random.default_rng(3), n_img=12)
    classes = ["a", "b", "c"]
    want = evaluate_dota(results, classes, pairwise=lambda A, B: np.array([opoly.iou_poly(x, y) for x, y in zip(A, B)]))
    got = evaluate_dota(results, classes, device=cuda)
    for k in want:
        assert got[k] == pytest.approx(want[k], abs=1e-12)


# ---------------------------------------------------------------- f4: in-model fp32 polygon NMS (ops/nms_poly.py:186-224)
@pytest.mark.parametrize("n,span,thr", [(1, 500, 0.1), (64, 300, 0.1), (65, 300, 0.3), (1000, 900, 0.1),
                                        (3000, 1024, 0.1), (700, 200, 0.5)])
def test_poly_nms_f32_vs_oracle(cuda, n, span, thr):
    """Bit-exact keep lists against the oracle's restatement of the reference's float arithmetic, and bit-identical
    IoU values on a sample of pairs (general quadrilaterals, both orientations, concave ones included)."""
    import oracle
    from rs_detection_amd.ops import poly_nms, poly_iou_f32
    rng = np.random.default_rng(n)
    q = _random_quads(rng, n, span, convex_only=False).astype(np.float32)
    dets = np.concatenate([q, rng.uniform(0.05, 1, (n, 1)).astype(np.float32)], 1)
    dets[::7, 8] = dets[0, 8]                           # score ties: index order decides (stable)
    c = oracle.c()
    keep = poly_nms(torch.from_numpy(dets).to(cuda), thr).cpu().numpy()
    want = c.poly_nms(dets, thr)
    assert len(keep) == len(want) and (keep == want).all()
    m = min(n, 128)
    got = poly_iou_f32(torch.from_numpy(q[:m]).to(cuda), torch.from_numpy(q[-m:]).to(cuda)).cpu().numpy()
    assert (got == c.poly_iou_f32(q[:m], q[-m:])).all()


def test_multiclass_poly_nms_and_edges(cuda):
    import oracle
    from rs_detection_amd.ops import multiclass_poly_nms, poly_nms
    rng = np.random.default_rng(5)
    n = 900
    q = _random_quads(rng, n, 800).astype(np.float32)
    scores = rng.uniform(0.05, 1, n).astype(np.float32)
    labels = rng.integers(0, 15, n)
    dets, lab = multiclass_poly_nms(torch.from_numpy(q).to(cuda), torch.from_numpy(scores).to(cuda),
                                    torch.from_numpy(labels).to(cuda), 0.1)
    # the same thing with the oracle: the offsets are computed in float32 exactly as nms_poly.py:213-216 does
    mc = np.float32(q.max() - q.min())
    off = labels.astype(np.float32) * (mc + np.float32(1))
    keep = oracle.c().poly_nms(np.concatenate([q + off[:, None], scores[:, None]], 1).astype(np.float32), 0.1)
    assert (lab.cpu().numpy() == labels[keep]).all()
    assert (dets.cpu().numpy() == np.concatenate([q[keep], scores[keep][:, None]], 1)).all()
    assert (np.diff(dets[:, 8].cpu().numpy()) <= 0).all()          # descending score, `order_t[keep]`
    # empty input, CPU tensors (no fallback), wrong width
    e, el = multiclass_poly_nms(torch.zeros(0, 8, device=cuda), torch.zeros(0, device=cuda),
                                torch.zeros(0, dtype=torch.int64, device=cuda), 0.1)
    assert tuple(e.shape) == (0, 9) and el.numel() == 0
    assert poly_nms(torch.zeros(0, 9, device=cuda), 0.1).numel() == 0
    with pytest.raises(Exception):
        poly_nms(torch.zeros(4, 9), 0.1)
    with pytest.raises(AssertionError):
        poly_nms(torch.zeros(4, 8, device=cuda), 0.1)


# ---------------------------------------------------------------- poly_iou_loss over convex_sort (models/losses/poly_iou_loss.py)
def test_poly_iou_loss_matches_polygon_iou_and_has_gradients(cuda):
    """IoU inside the loss == the fp64 polygon-IoU kernel (independent algorithm: Sutherland-Hodgman) to 1e-3 on
    rotated boxes that overlap; closed forms; gradients flow to the predicted boxes; registry builds the modules."""
    from rs_detection_amd.models.losses.poly_iou_loss import (poly_overlaps, poly_iou_loss, poly_giou_loss,
                                                               PolyIoULoss, PolyGIoULoss, convex_areas, poly_enclose)
    from rs_detection_amd.ops import poly_iou_matrix
    from rs_detection_amd.ops.bbox_transforms import obb2poly
    from rs_detection_amd.utils.registry import LOSSES, build_from_cfg
    rng = np.random.default_rng(11)
    n = 600
    c = rng.uniform(100, 900, (n, 2))
    t = np.concatenate([c, rng.uniform(20, 120, (n, 2)), rng.uniform(-np.pi / 2, np.pi / 2, (n, 1))], 1)
    p = t + np.concatenate([rng.normal(0, 8, (n, 2)), rng.normal(0, 6, (n, 2)), rng.normal(0, 0.3, (n, 1))], 1)
    p[:, 2:4] = np.maximum(p[:, 2:4], 5)
    pred = torch.tensor(p, dtype=torch.float32, device=cuda, requires_grad=True)
    tgt = torch.tensor(t, dtype=torch.float32, device=cuda)
    ious, union, _, _ = poly_overlaps(pred, tgt)
    want = torch.diagonal(poly_iou_matrix(obb2poly(pred.detach()).double(), obb2poly(tgt).double())).float()
    assert float((ious.detach() - want).abs().max()) <= 2e-3, float((ious.detach() - want).abs().max())
    loss = poly_iou_loss(pred, tgt)
    loss.backward()
    assert torch.isfinite(pred.grad).all() and float(pred.grad.abs().sum()) > 0
    # closed forms: identical boxes -> IoU 1, loss ~ 0; unit squares shifted by half a side -> IoU 1/3; disjoint -> eps
    sq = torch.tensor([[0., 0., 2., 2.], [0., 0., 2., 2.], [0., 0., 2., 2.]], device=cuda)       # hbb layout
    other = torch.tensor([[0., 0., 2., 2.], [1., 0., 3., 2.], [10., 10., 12., 12.]], device=cuda)
    i2, _, _, _ = poly_overlaps(sq, other)
    assert torch.allclose(i2, torch.tensor([1., 1 / 3., 0.], device=cuda), atol=1e-5)
    l_lin = poly_iou_loss(sq, other, linear=True, reduction='none')
    assert torch.allclose(l_lin, torch.tensor([0., 2 / 3., 1.], device=cuda), atol=1e-5)
    # GIoU: disjoint squares 2x2 at distance 10: enclosing hull area known
    g = poly_giou_loss(sq, other, reduction='none')
    pts, m = poly_enclose(sq.new_tensor([[[0, 0], [2, 0], [2, 2], [0, 2]]]), sq.new_tensor([[[10, 10], [12, 10], [12, 12], [10, 12]]]))
    hull = float(convex_areas(pts, m)[0])
    assert abs(hull - 44.0) < 1e-3            # hexagon (0,0) (2,0) (12,10) (12,12) (10,12) (0,2): shoelace = 44
    assert abs(float(g[2]) - (1 - (0 - (hull - 8) / hull))) < 1e-4
    # modules through the registry, weights, reductions
    m1 = build_from_cfg(dict(type='PolyIoULoss', linear=True, loss_weight=2.0), LOSSES)
    m2 = build_from_cfg(dict(type='PolyGIoULoss'), LOSSES)
    assert isinstance(m1, PolyIoULoss) and isinstance(m2, PolyGIoULoss)
    w = torch.tensor([1., 0., 1.], device=cuda)
    assert abs(float(m1(sq, other, weight=w, avg_factor=2.0)) - 2.0 * (0 + 0 + 1) / 2.0) < 1e-5
    assert float(m2(sq, other, weight=torch.zeros(3, device=cuda))) == 0.0
    assert float(m2(sq, other, weight=w, reduction_override='sum')) > 0
