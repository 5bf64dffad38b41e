"""SURVEY 8f rank 1, host side (no GPU): tile-name parsing / merge / DOTA Task-1 lines, voc_ap, voc_eval_dota and the
per-class driver against the line-by-line oracle restatements (oracle/poly.py), with the oracle's polygon IoU plugged
in where the product uses the GPU kernel.  Known answers for the polygon IoU oracle itself (shapely is absent: the
oracle's parity with it is unpinned, its geometry is pinned here by hand-computed areas)."""
import os

import numpy as np
import pytest

from oracle import poly as opoly
from rs_detection_amd.data.devkits import (parse_tile_name, poly2origpoly, merge_detections, mergesingle, voc_ap,
                                           voc_eval_dota, evaluate_dota)


def _sq(x, y, s):
    return [x, y, x + s, y, x + s, y + s, x, y + s]


def _rbox_poly(cx, cy, w, h, a):
    c, s = np.cos(a), np.sin(a)
    dx, dy = np.array([-w, w, w, -w]) / 2, np.array([-h, -h, h, h]) / 2
    return np.stack([cx + dx * c - dy * s, cy + dx * s + dy * c], 1).reshape(-1)


def test_oracle_iou_poly_known_answers():
    assert opoly.iou_poly(_sq(0, 0, 10), _sq(5, 0, 10)) == pytest.approx(50 / 150)
    assert opoly.iou_poly(_sq(0, 0, 10), _sq(0, 0, 10)) == pytest.approx(1.0)
    assert opoly.iou_poly(_sq(0, 0, 10), _sq(20, 20, 5)) == 0.0
    assert opoly.iou_poly(_sq(0, 0, 10), [5, -5, 15, 5, 5, 15, -5, 5]) == pytest.approx(0.5)        # square in a diamond
    cw = [0, 0, 0, 10, 10, 10, 10, 0]                                                              # clockwise input
    assert opoly.iou_poly(cw, _sq(5, 0, 10)) == pytest.approx(50 / 150)
    assert opoly.iou_poly(_sq(0, 0, 0.05), _sq(0, 0, 0.05)) == pytest.approx(0.0025 / 0.01)        # max(union, 0.01)
    # two unit-area-100 squares, one rotated by 45 degrees about the common centre: regular octagon
    a, b = _rbox_poly(0, 0, 10, 10, 0), _rbox_poly(0, 0, 10, 10, np.pi / 4)
    inter = 100 * 2 * (np.sqrt(2) - 1)
    assert opoly.iou_poly(a, b) == pytest.approx(inter / (200 - inter), rel=1e-12)
    # concave subject (an arrow head) clipped by a convex quad: picks the convex one as the clipper
    arrow = [0, 0, 10, 5, 0, 10, 4, 5]
    assert opoly.iou_poly(arrow, _sq(0, 0, 10)) == pytest.approx(30 / 100)
    assert opoly.iou_poly(_sq(0, 0, 10), arrow) == pytest.approx(30 / 100)


def test_voc_ap_known_values():
    rec, prec = np.array([0.5, 1.0]), np.array([1.0, 0.5])
    assert voc_ap(rec, prec) == pytest.approx(0.75)
    assert voc_ap(rec, prec, use_07_metric=True) == pytest.approx((6 * 1.0 + 5 * 0.5) / 11)
    assert voc_ap(np.array([0.0]), np.array([0.0])) == 0.0
    assert voc_ap(rec, prec) == opoly.voc_ap(rec, prec)


def test_tile_name_parsing_and_poly_back_projection():
    assert parse_tile_name("P0706__1__824___1648") == ("P0706", 824, 1648, 1.0)
    assert parse_tile_name("P1234__0.5__0___512") == ("P1234", 0, 512, 0.5)
    assert poly2origpoly([1, 2, 3, 4, 5, 6, 7, 8], 100, 200, 0.5) == [202.0, 404.0, 206.0, 408.0, 210.0, 412.0, 214.0, 416.0]


def test_merge_detections_and_task1_file_format(tmp_path):
    # the same object seen in two overlapping tiles + one unrelated object, 2 original images
    base = _rbox_poly(900, 300, 60, 20, 0.3)
    def tile_line(name, x, y, score, poly):
        return ("%s__1__%d___%d" % (name, x, y), "%.4f" % score, ["%.4f" % v for v in poly - np.tile([x, y], 4)])
    lines = [tile_line("P1", 0, 0, 0.9, base), tile_line("P1", 824, 0, 0.8, base + 1.0),
             tile_line("P1", 824, 0, 0.7, _rbox_poly(1500, 300, 40, 40, 0)), tile_line("P2", 0, 824, 0.6, base)]
    merged = merge_detections(lines, 0.1, lambda d, t: opoly.py_cpu_nms_poly_fast(d, t))
    assert set(merged) == {"P1", "P2"} and len(merged["P1"]) == 2 and len(merged["P2"]) == 1
    assert [d[-1] for d in merged["P1"]] == [0.9, 0.7]                      # duplicate from the second tile suppressed
    np.testing.assert_allclose(merged["P1"][0][:8], base, atol=1e-3)
    src, dst = tmp_path / "before_nms", tmp_path / "after_nms"
    src.mkdir(), dst.mkdir()
    with open(src / "Ship.txt", "w") as f:
        for name, score, poly in lines:
            f.write(" ".join([name, score] + poly) + "\n")
    mergesingle(str(dst), lambda d, t: opoly.py_cpu_nms_poly_fast(d, t), str(src / "Ship.txt"))
    out = [l.split() for l in open(dst / "Ship.txt").read().strip().splitlines()]
    assert [l[0] for l in out] == ["P1", "P1", "P2"] and all(len(l) == 10 for l in out)
    assert float(out[0][1]) == 0.9


def _synthetic_eval_set(rng, n_img=6, n_cls=3):
    results = []
    for _ in range(n_img):
        k = int(rng.integers(2, 8))
        gts = np.stack([_rbox_poly(*rng.uniform(100, 900, 2), rng.uniform(20, 120), rng.uniform(10, 60), rng.uniform(-1.5, 1.5))
                        for _ in range(k)])
        labels = rng.integers(1, n_cls + 1, k)
        ignore = np.stack([_rbox_poly(*rng.uniform(100, 900, 2), 50, 20, 0.2)]) if rng.random() < 0.5 else np.zeros((0, 8))
        det_p, det_s, det_l = [], [], []
        for g, l in zip(gts, labels):                      # noisy copies, some duplicates, some misses
            for _ in range(int(rng.integers(0, 3))):
                det_p.append(g + rng.normal(0, 3, 8))
                det_s.append(rng.uniform(0.3, 1))
                det_l.append(l - 1 if rng.random() < 0.9 else int(rng.integers(0, n_cls)))
        for _ in range(int(rng.integers(0, 4))):           # false positives
            det_p.append(_rbox_poly(*rng.uniform(100, 900, 2), 40, 20, rng.uniform(-1, 1)))
            det_s.append(rng.uniform(0.05, 0.6))
            det_l.append(int(rng.integers(0, n_cls)))
        res = (np.array(det_p).reshape(-1, 8), np.array(det_s), np.array(det_l, dtype=np.int64))
        results.append((res, dict(polys=gts * 2.0, labels=labels, polys_ignore=ignore * 2.0, scale_factor=2.0)))
    return results


def test_voc_eval_dota_and_driver_against_literal_restatement():
    rng = np.random.default_rng(3)
    results = _synthetic_eval_set(rng)
    classes = ["a", "b", "c"]
    pair = lambda A, B: np.array([opoly.iou_poly(x, y) for x, y in zip(A, B)])
    aps = evaluate_dota(results, classes, pairwise=pair)
    assert set(aps) == {"eval/1_a_AP", "eval/2_b_AP", "eval/3_c_AP", "eval/0_meanAP"}
    assert 0 < aps["eval/0_meanAP"] <= 1
    # literal per-class loop (dota.py:113-141 + voc_eval.py:236-336) on the same data
    dets, gts, diff = [], [], {}
    for i, (res, tg) in enumerate(results):
        p, s, l = res
        if p.size:
            dets.append(np.concatenate([np.full((len(s), 1), i), p, s[:, None], (l + 1)[:, None]], 1))
        g = tg["polys"] / tg["scale_factor"]
        gts.append(np.concatenate([np.full((len(g), 1), i), g, tg["labels"][:, None]], 1))
        diff[i] = tg["polys_ignore"] / tg["scale_factor"]
    dets, gts = np.concatenate(dets), np.concatenate(gts)
    want = []
    for c in range(1, 4):
        cd, cg = dets[dets[:, -1] == c][:, :-1], gts[gts[:, -1] == c][:, :-1]
        cls_gts = {}
        for idx in np.unique(gts[:, 0]):
            g = cg[cg[:, 0] == idx][:, 1:]
            dg = diff[int(idx)].reshape(-1, 8)
            d = np.zeros(len(g) + len(dg), bool)
            d[len(g):] = True
            cls_gts[int(idx)] = dict(box=np.concatenate([g, dg]), det=[False] * (len(g) + len(dg)), difficult=d)
        want.append(opoly.voc_eval_dota(cd, cls_gts)[2])
    got = [aps["eval/%d_%s_AP" % (i + 1, c)] for i, c in enumerate(classes)]
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    assert aps["eval/0_meanAP"] == pytest.approx(sum(want) / 3)
    # signature parity: an explicit iou_func is honoured
    cd = dets[dets[:, -1] == 1][:, :-1]
    g1 = {0: dict(box=np.zeros((0, 8)), det=[], difficult=np.zeros(0, bool))}
    assert voc_eval_dota(cd[:0], g1, iou_func=opoly.iou_poly) == (0., 0., 0.)


# ---- test-time submission pipeline (data/devkits/data_merge.py) and poly2obb -------------------------------------
def test_flip_box_prepare_data_and_fair_csv(tmp_path):
    """data_merge.py:14-48 and dota_to_fair.py:6-36,102-118 restated: a flipped tile's polygon is mirrored back with the
    tile's own size; one Task-1 line per detection with 4 decimals; the FAIR1M-1.5 csv carries `<id>.tif,class,...`."""
    from rs_detection_amd.data.devkits import data_merge as DM
    box = [10., 20., 30., 20., 30., 40., 10., 40.]
    tgt = dict(ori_img_size=(100, 80), img_file="/x/P0007__1.0__0___0.png")
    assert DM.flip_box(box, tgt) == box
    assert DM.flip_box(box, dict(tgt, flip_mode="H")) == [90., 20., 70., 20., 70., 40., 90., 40.]
    assert DM.flip_box(box, dict(tgt, flip_mode="V")) == [10., 60., 30., 60., 30., 40., 10., 40.]
    assert DM.flip_box(box, dict(tgt, flip_mode="HV")) == [90., 60., 70., 60., 70., 40., 90., 40.]
    res = [((np.array([box, box]), np.array([0.91234, 0.5]), np.array([0, 2])), tgt),
           ((np.array([box]), np.array([0.7]), np.array([0])), dict(tgt, flip_mode="H"))]
    classes = ["Airplane", "Ship", "Vehicle"]
    DM.prepare_data(res, str(tmp_path / "before"), classes)
    air = open(tmp_path / "before" / "Airplane.txt").read().splitlines()
    assert air[0] == "P0007__1.0__0___0 0.9123 10.0000 20.0000 30.0000 20.0000 30.0000 40.0000 10.0000 40.0000"
    assert air[1].startswith("P0007__1.0__0___0 0.7000 90.0000 20.0000") and len(air) == 2
    assert len(open(tmp_path / "before" / "Vehicle.txt").read().splitlines()) == 1
    # merged files -> csv
    (tmp_path / "imgs").mkdir()
    for n in ("P0007__1.0__0___0.png", "P0012__1.0__0___0.png", "notes.txt"):
        (tmp_path / "imgs" / n).write_text("")
    (tmp_path / "after").mkdir()
    (tmp_path / "after" / "Tennis_Court.txt").write_text("P0007 0.9 1.0 2.0 3.0 4.0 5.0 6.0 7.0 8.0\n")
    got = DM.pick_res(str(tmp_path / "after"), str(tmp_path / "imgs"), keep_underline=True)
    assert set(got) == {"P0007", "P0012"} and got["P0012"] == [] and got["P0007"][0]["cls"] == "Tennis_Court"
    assert DM.pick_res(str(tmp_path / "after"), str(tmp_path / "imgs"))["P0007"][0]["cls"] == "Tennis Court"
    csv = DM.dota_to_fair1m_1_5(str(tmp_path / "after"), str(tmp_path / "fair"), str(tmp_path / "imgs"), "sub")
    assert open(csv).read() == "7.tif,Tennis_Court,1.0000,2.0000,3.0000,4.0000,5.0000,6.0000,7.0000,8.0000,0.9000\n"
    # dataset_type 'FAIR' (dota_to_fair.py:37-100): one XML per image of images_dir, class names with spaces, the first
    # corner repeated to close the rectangle; an unknown type is refused BEFORE any inference (ImageDataset.__init__)
    import xml.etree.ElementTree as ET
    DM.dota_to_fair(str(tmp_path / "after"), str(tmp_path / "fairxml"), str(tmp_path / "imgs"))
    assert sorted(os.listdir(tmp_path / "fairxml")) == ["12.xml", "7.xml"]
    root = ET.parse(tmp_path / "fairxml" / "7.xml").getroot()
    assert root.find("source/filename").text == "7.tif" and root.find("size/width").text == "1000"
    obj = root.findall("objects/object")
    assert len(obj) == 1 and obj[0].find("possibleresult/name").text == "Tennis Court"
    assert obj[0].find("possibleresult/probability").text == "0.9"
    assert [p.text for p in obj[0].findall("points/point")] == ["1.0, 2.0", "3.0, 4.0", "5.0, 6.0", "7.0, 8.0", "1.0, 2.0"]
    assert ET.parse(tmp_path / "fairxml" / "12.xml").getroot().findall("objects/object") == []
    with pytest.raises(ValueError):
        DM.check_dataset_type("COCO")
    from rs_detection_amd.data.image import ImageDataset
    with pytest.raises(ValueError):
        ImageDataset(images_dir=str(tmp_path / "imgs"), dataset_type="FAIR2")
    assert ImageDataset(images_dir=str(tmp_path / "imgs"), dataset_type="FAIR").dataset_type == "FAIR"


def test_poly2obb_min_area_rect():
    """The role of cv2.minAreaRect (absent here): exact on rectangles (any point order, duplicates), the enclosing
    rectangle of a general quadrilateral has the smallest area over all hull-edge directions, w >= h and the angle in
    [-pi/2, pi/2) -- the regular form `bbox2type(polys, 'obb')` promises (bbox_transforms.py:547-575)."""
    import torch
    from rs_detection_amd.ops import bbox_transforms as T
    rng = np.random.default_rng(0)
    obb = np.stack([rng.uniform(50, 200, 60), rng.uniform(50, 200, 60), rng.uniform(20, 80, 60), rng.uniform(5, 19, 60),
                    rng.uniform(-np.pi / 2, np.pi / 2 - 1e-3, 60)], 1).astype(np.float32)
    poly = T.obb2poly(torch.from_numpy(obb))
    np.testing.assert_allclose(T.poly2obb(poly).numpy(), obb, atol=1e-3)
    perm = poly.reshape(-1, 4, 2)[:, [2, 0, 3, 1]].reshape(-1, 8)                 # another vertex order
    np.testing.assert_allclose(T.bbox2type(perm, 'obb').numpy(), obb, atol=1e-3)
    quad = np.array([[0, 0, 10, 1, 12, 6, 1, 4]], np.float32)
    x, y, w, h, t = T.poly2obb(torch.from_numpy(quad))[0].tolist()
    assert w >= h and -np.pi / 2 <= t < np.pi / 2
    pts = quad.reshape(4, 2)
    for ang in np.linspace(0, np.pi, 721):                                          # no direction does better
        u = np.array([np.cos(ang), np.sin(ang)])
        v = np.array([-u[1], u[0]])
        assert np.ptp(pts @ u) * np.ptp(pts @ v) >= w * h - 1e-3
    rect = T.obb2poly(torch.tensor([[x, y, w, h, t]])).numpy().reshape(4, 2)        # ... and it contains the points
    e = np.roll(rect, -1, 0) - rect
    for q in pts:
        s = np.sign(e[:, 0] * (q - rect)[:, 1] - e[:, 1] * (q - rect)[:, 0])
        assert (s >= -1e-6).all() or (s <= 1e-6).all() or np.abs(e[:, 0] * (q - rect)[:, 1] - e[:, 1] * (q - rect)[:, 0]).min() < 1e-2
    assert T.poly2obb(torch.zeros((0, 8))).shape == (0, 5)
