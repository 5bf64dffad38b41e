"""CPU: the DOTA input pipeline against outputs of THE REFERENCE'S OWN transforms code.

tests/golden/transforms.npz was produced in the build container by running /root/reference/python/jdet/data/
transforms.py:190-823 and models/boxes/box_ops.py on seeded images / targets (tests/golden/make_transforms_golden.py);
only the arrays travel.  Same seeds here -> the same random decisions (both sides draw from Python's ``random`` in the
same order), so pixels must agree exactly and boxes to float round-off."""
import copy
import os
import random

import numpy as np
import pytest
from PIL import Image

from rs_detection_amd.data import box_np, transforms as T

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "transforms.npz"), allow_pickle=False)
NORM = dict(type="Normalize", mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_bgr=False)


def test_box_helpers_equal_the_reference():
    rb = G["boxes_in"]
    np.testing.assert_allclose(box_np.norm_angle_np(G["norm_angle_in"].copy(), 'le135'), G["norm_angle_le135"], atol=1e-6)
    np.testing.assert_allclose(box_np.norm_angle_np(G["norm_angle_in"].copy(), 'le90'), G["norm_angle_le90"], atol=1e-6)
    for ver in ("le135", "le90"):
        polys = box_np.rotated_box_to_poly_np(rb.copy(), ver)
        np.testing.assert_allclose(polys, G["r2p_" + ver], rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose(box_np.poly_to_rotated_box_np(G["r2p_" + ver].copy(), ver), G["p2r_" + ver],
                                   rtol=1e-5, atol=1e-4)
    hb, pl = box_np.rotated_box_to_bbox_np(rb.copy())
    np.testing.assert_allclose(hb, G["r2bbox_h"], rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(pl, G["r2bbox_p"], rtol=1e-6, atol=1e-4)


def _seed_of(name):
    fixed = {"resize_le135": 1, "resize_le90": 2, "flip_h": 3, "flip_v": 4, "pad": 5, "normalize": 6,
             "compose_s2anet": 7, "compose_cfg4": 11}
    if name in fixed:
        return fixed[name]
    k = int(name[-1])                                   # ra90_k: the first seed whose draw gives k quarter turns
    s, seen = 0, set()
    while True:
        random.seed(s)
        d = int(random.random() * 100) // 25
        if d not in seen:
            seen.add(d)
            if d == k:
                return s
        s += 1


CASES = {
    "resize_le135": lambda: T.RotatedResize(128, 128),
    "resize_le90": lambda: T.RotatedResize(150, 256, angle_version='le90'),
    "flip_h": lambda: T.RotatedRandomFlip(prob=1.0),
    "flip_v": lambda: T.RotatedRandomFlip(prob=1.0, direction="vertical"),
    "ra90_0": lambda: T.RandomRotateAug(random_rotate_on=True), "ra90_1": lambda: T.RandomRotateAug(random_rotate_on=True),
    "ra90_2": lambda: T.RandomRotateAug(random_rotate_on=True), "ra90_3": lambda: T.RandomRotateAug(random_rotate_on=True),
    "pad": lambda: T.Pad(size_divisor=32),
    "normalize": lambda: T.Normalize(mean=NORM["mean"], std=NORM["std"], to_bgr=False),
    "compose_s2anet": lambda: T.Compose([dict(type="RotatedResize", min_size=128, max_size=128),
                                         dict(type="RotatedRandomFlip", prob=0.5), dict(type="Pad", size_divisor=32), NORM]),
    "compose_cfg4": lambda: T.Compose([dict(type="RotatedResize", min_size=128, max_size=128),
                                       dict(type="RotatedRandomFlip", prob=0.5),
                                       dict(type="RandomRotateAug", random_rotate_on=True),
                                       dict(type="Pad", size_divisor=32), NORM]),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_transform_equals_the_reference_output(name):
    img = G[name + "/img_in"]
    h, w = img.shape[:2]
    tgt = dict(rboxes=G[name + "/in_rboxes"].copy(), hboxes=G[name + "/in_hboxes"].copy(),
               polys=G[name + "/in_polys"].copy(), labels=G[name + "/in_labels"].copy(),
               rboxes_ignore=G[name + "/in_rboxes_ignore"].copy(), img_size=(w, h), ori_img_size=(w, h), scale_factor=1.0)
    random.seed(_seed_of(name))
    im2, t2 = CASES[name]()(Image.fromarray(img), copy.deepcopy(tgt))
    got = np.asarray(im2) if isinstance(im2, Image.Image) else im2
    want = G[name + "/img_out"]
    assert got.shape == want.shape, (got.shape, want.shape)
    if want.dtype == np.uint8:
        assert (got == want).all()                                  # PIL resize / flip / rotate / paste: same pixels
    else:
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-5)
    for k in ("rboxes", "hboxes", "polys", "rboxes_ignore"):
        np.testing.assert_allclose(np.asarray(t2[k]), G[name + "/out_" + k], rtol=1e-5, atol=2e-3, err_msg=k)
    meta = G[name + "/meta"]
    assert tuple(t2["img_size"]) == (int(meta[0]), int(meta[1]))
    assert abs(float(t2.get("scale_factor", 1.0)) - meta[2]) < 1e-9
    assert tuple(t2.get("pad_shape", t2["img_size"])) == (int(meta[3]), int(meta[4]))
    assert float(t2.get("rotate_angle", -1)) == meta[5]
    assert {"horizontal": 1, "vertical": 2}.get(t2.get("flip"), 0) == int(meta[6])
