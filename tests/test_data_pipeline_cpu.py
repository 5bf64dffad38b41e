"""SURVEY 8f rank 2 (host side): NumPy box helpers, PIL transforms, CustomDataset / DOTADataset on a generated
labels.pkl + images folder -- properties and NumPy twins (reference: data/transforms.py, data/custom.py, data/dota.py,
models/boxes/box_ops.py:440-665)."""
import os
import pickle
import random

import numpy as np
import pytest
from PIL import Image

import rs_detection_amd.data  # noqa: F401
from rs_detection_amd.data import box_np
from rs_detection_amd.data.transforms import (Compose, RandomRotateAug, RotatedResize, RotatedRandomFlip, Pad, Normalize)
from rs_detection_amd.utils.registry import DATASETS, TRANSFORMS, build_from_cfg
from oracle import poly as opoly


def _rboxes(rng, n, w=200, h=160):
    bw = rng.uniform(10, 60, n)
    bh = rng.uniform(4, np.minimum(bw, 30))
    return np.stack([rng.uniform(40, w - 40, n), rng.uniform(40, h - 40, n), bw, bh,
                     rng.uniform(-np.pi / 4, 3 * np.pi / 4, n)], 1).astype(np.float32)


def _target(rb, size):
    hb, polys = box_np.rotated_box_to_bbox_np(rb)
    return dict(rboxes=rb.copy(), hboxes=hb.astype(np.float32), polys=polys.astype(np.float32),
                labels=np.ones(len(rb), np.int32), rboxes_ignore=np.zeros((0, 5), np.float32),
                hboxes_ignore=np.zeros((0, 4)), polys_ignore=np.zeros((0, 8)), img_size=size, ori_img_size=size,
                scale_factor=1.0)


def test_poly_rbox_roundtrip_and_begin_point():
    rng = np.random.default_rng(0)
    rb = _rboxes(rng, 50)
    polys = box_np.rotated_box_to_poly_np(rb, 'le135')
    back = box_np.poly_to_rotated_box_np(polys, 'le135')
    np.testing.assert_allclose(back[:, :4], rb[:, :4], atol=2e-3)
    d = np.abs(back[:, 4] - rb[:, 4])
    assert np.minimum(d, np.pi - d).max() < 1e-3 and (back[:, 4] >= -np.pi / 4 - 1e-6).all() and (back[:, 4] < 3 * np.pi / 4 + 1e-6).all()
    # the polygon is the same set of corners, started at the vertex nearest the hull's top-left corner
    sq = np.array([10., 0., 10., 10., 0., 10., 0., 0.])          # starts at the top-right corner
    np.testing.assert_allclose(box_np.get_best_begin_point_single(sq), [0., 0., 10., 0., 10., 10., 0., 10.])
    hb, p = box_np.rotated_box_to_bbox_np(np.array([[10., 20., 8., 4., np.pi / 2]], np.float32))
    np.testing.assert_allclose(hb, [[8., 16., 12., 24.]], atol=1e-5)
    assert box_np.rotated_box_to_bbox_np(np.zeros((0, 5)))[0].shape == (0, 4)
    assert box_np.norm_angle_np(np.pi) == pytest.approx(0.0) and box_np.norm_angle_np(-np.pi / 2) == pytest.approx(np.pi / 2)


def test_flip_is_an_involution_and_matches_geometry():
    rng = np.random.default_rng(1)
    size = (200, 160)
    img = Image.fromarray(rng.integers(0, 255, (160, 200, 3), dtype=np.uint8))
    for direction in ("horizontal", "vertical"):
        rb = _rboxes(rng, 20)
        t = _target(rb, size)
        flip = RotatedRandomFlip(prob=1.0, direction=direction)
        img1, t1 = flip(img, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in t.items()})
        assert t1["flip"] == direction
        # geometry: the flipped rotated box covers the mirrored polygon (IoU 1 against the mirrored corners)
        mirrored = t["polys"].copy()
        if direction == "horizontal":
            mirrored[:, 0::2] = 200 - mirrored[:, 0::2] - 1
        else:
            mirrored[:, 1::2] = 160 - mirrored[:, 1::2] - 1
        np.testing.assert_allclose(t1["polys"], mirrored, atol=1e-4)
        got = box_np.rotated_box_to_poly_np(t1["rboxes"], 'le135')
        for a, b in zip(got, mirrored):
            assert opoly.iou_poly(a, b) > 0.999
        img2, t2 = flip(img1, t1)
        np.testing.assert_allclose(t2["rboxes"][:, :4], rb[:, :4], atol=1e-4)
        np.testing.assert_allclose(t2["hboxes"], t["hboxes"], atol=1e-4)
        assert np.array_equal(np.array(img2), np.array(img))


def test_rotate_aug_four_quarter_turns_are_identity():
    rng = np.random.default_rng(2)
    img = Image.fromarray(rng.integers(0, 255, (160, 200, 3), dtype=np.uint8))
    rb = _rboxes(rng, 15)
    t = _target(rb, (200, 160))
    aug = RandomRotateAug(angle_version='le135', random_rotate_on=True)
    cur_img, cur = img, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in t.items()}
    for _ in range(4):
        aug._rotate_boxes_90(cur, cur_img.size)
        cur_img = cur_img.rotate(90, expand=True)
    assert np.array_equal(np.array(cur_img), np.array(img))
    np.testing.assert_allclose(cur["polys"], t["polys"], atol=1e-3)
    np.testing.assert_allclose(cur["hboxes"], t["hboxes"], atol=1e-3)
    for a, b in zip(box_np.rotated_box_to_poly_np(cur["rboxes"], 'le135'), t["polys"]):
        assert opoly.iou_poly(a, b) > 0.999
    random.seed(3)
    out_img, out = aug(img, dict(t))
    assert out["rotate_angle"] in (0, 90, 180, 270) and out_img.size in ((200, 160), (160, 200))


def test_resize_pad_normalize():
    rng = np.random.default_rng(4)
    img = Image.fromarray(rng.integers(0, 255, (160, 200, 3), dtype=np.uint8))
    t = _target(_rboxes(rng, 10), (200, 160))
    img2, t2 = RotatedResize(min_size=320, max_size=400)(img, {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in t.items()})
    # transforms.py:424-429 clips the target short side to [short/1.5, short*1.5]: 320 -> 240 for a 160-px side
    assert img2.size == (300, 240) and t2["scale_factor"] == 1.5 and t2["img_size"] == (300, 240) and t2["keep_ratio"]
    np.testing.assert_allclose(t2["rboxes"][:, :4], t["rboxes"][:, :4] * 1.5, rtol=2e-3, atol=2e-2)
    np.testing.assert_allclose(t2["hboxes"], t["hboxes"] * 1.5, atol=1e-3)
    img3, t3 = Pad(size_divisor=128)(img2, t2)
    assert img3.size == (384, 256) and t3["pad_shape"] == (384, 256)
    assert np.array(img3)[245:, :, :].max() == 0
    arr, t4 = Normalize(mean=[1., 2., 3.], std=[2., 4., 8.], to_bgr=True)(img3, t3)
    raw = np.array(img3).transpose(2, 0, 1)[::-1].astype(np.float32)
    np.testing.assert_allclose(arr, (raw - np.float32([1, 2, 3]).reshape(3, 1, 1)) / np.float32([2, 4, 8]).reshape(3, 1, 1))
    assert t4["to_bgr"] is True and arr.shape == (3, 256, 384)


def _make_dataset(tmp_path, n=7, seed=5):
    rng = np.random.default_rng(seed)
    (tmp_path / "images").mkdir()
    infos = []
    for i in range(n):
        w, h = (200, 160) if i % 2 == 0 else (160, 200)
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(tmp_path / "images" / ("P%04d.png" % i))
        k = 0 if i == 3 else int(rng.integers(2, 6))
        infos.append(dict(filename="P%04d.png" % i, width=w, height=h,
                          ann=dict(bboxes=_rboxes(rng, k, w, h), labels=rng.integers(1, 16, k).astype(np.int64),
                                   bboxes_ignore=np.zeros((0, 5), np.float32), labels_ignore=np.zeros((0,), np.int64))))
    with open(tmp_path / "labels.pkl", "wb") as f:
        pickle.dump(infos, f)
    return infos


def test_dota_dataset_batches_shards_and_evaluates(tmp_path):
    infos = _make_dataset(tmp_path)
    cfg = dict(type="DOTADataset", dataset_dir=str(tmp_path), batch_size=2, shuffle=True, drop_last=False,
               transforms=[dict(type="RotatedResize", min_size=256, max_size=256),
                           dict(type="RotatedRandomFlip", prob=0.5), dict(type="Pad", size_divisor=32),
                           dict(type="Normalize", mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_bgr=False)])
    ds = build_from_cfg(cfg, DATASETS)
    assert len(ds) == 6 and len(ds.CLASSES) == 15          # the empty image is filtered (filter_empty_gt)
    random.seed(0)
    seen = []
    for images, targets in ds:
        assert images.dtype == np.float32 and images.shape[1] == 3 and images.shape[2] % 32 == 0 and images.shape[3] % 32 == 0
        for t in targets:
            assert set(["rboxes", "hboxes", "polys", "labels", "img_size", "pad_shape", "scale_factor", "img_file"]) <= set(t)
            assert t["rboxes"].dtype == np.float32 and t["labels"].dtype == np.int32 and t["rboxes"].shape[1] == 5
            assert max(t["img_size"]) <= 256
            seen.append(t["filename"])
    assert sorted(seen) == sorted(i["filename"] for i in infos if len(i["ann"]["bboxes"]))
    # two ranks: disjoint halves of the same shuffled stream, equal length
    parts = []
    for r in range(2):
        d = build_from_cfg(cfg, DATASETS)
        d.set_shard(r, 2)
        parts.append([t["filename"] for _, ts in d for t in ts])
    assert len(parts[0]) == len(parts[1]) == 2 and not set(parts[0]) & set(parts[1])
    # balance_category only repeats images
    bal = build_from_cfg(dict(cfg, balance_category=True), DATASETS)
    assert len(bal) >= len(ds)
    # evaluate: the ground truth as detections scores mAP 1 on every class that occurs
    plain = build_from_cfg(dict(type="DOTADataset", dataset_dir=str(tmp_path), transforms=[]), DATASETS)
    results = []
    for i in range(len(plain)):
        _, t = plain[i]
        t = dict(t, polys=t["polys"].astype(np.float64))
        results.append(((t["polys"].copy(), np.full(len(t["labels"]), 0.9), t["labels"] - 1), t))
    aps = plain.evaluate(results, pairwise=lambda A, B: np.array([opoly.iou_poly(a, b) for a, b in zip(A, B)]))
    present = {int(l) for _, t in results for l in t["labels"]}
    for c in present:
        assert aps["eval/%d_%s_AP" % (c, plain.CLASSES[c - 1])] == pytest.approx(1.0)
    # Task-1 files
    plain.parse_result([((np.concatenate([results[0][1]["rboxes"], np.full((len(results[0][1]["labels"]), 1), 0.5)], 1),
                          results[0][1]["labels"] - 1), "P0000.png")], str(tmp_path / "out"))
    files = os.listdir(tmp_path / "out")
    assert files and all(len(l.split()) == 10 for f in files for l in open(tmp_path / "out" / f))
    assert "Compose" in TRANSFORMS._modules if hasattr(TRANSFORMS, "_modules") else True
