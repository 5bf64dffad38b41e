"""Generates tests/golden/transforms.npz by RUNNING THE REFERENCE'S OWN data pipeline code (build container only:
/root/reference must be mounted; the fixture -- plain arrays -- is what travels).

The reference's jdet/data/transforms.py and jdet/models/boxes/box_ops.py are imported from where they lie.  Their
module-level `import jittor / cv2 / skimage` cannot be satisfied here (not installed, no network); none of the classes
exercised below touches them (PIL + NumPy only), so those three names are bound to inert placeholder modules for the
import, and the jdet package __init__ files (which pull in every Jittor model) are bypassed by registering bare
package modules whose __path__ points at the reference directories.  No reference source is copied or modified.

Cases: RotatedResize (+ clip), RotatedRandomFlip horizontal / vertical, RandomRotateAug for each of the four quarter
turns, Pad, Normalize, the S2ANet train Compose and the config[4] Compose (flip + ra90), on seeded images / targets;
plus the NumPy box helpers norm_angle, rotated_box_to_poly_np, poly_to_rotated_box_np, rotated_box_to_bbox_np."""
import copy
import importlib
import os
import random
import sys
import types

import numpy as np
from PIL import Image

REF = "/root/reference/python/jdet"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    class _Inert(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)

            class _C:
                def __init__(self, *a, **k):
                    pass

                def __call__(self, *a, **k):
                    return a[0] if a and callable(a[0]) else self

                def __getattr__(self, n):
                    return _C()
            _C.__name__ = name
            setattr(self, name, _C)
            return _C
    for m in ("jittor", "jittor.nn", "jittor.dataset", "cv2", "skimage", "pycocotools", "pycocotools.coco", "shapely",
              "shapely.geometry"):
        sys.modules.setdefault(m, _Inert(m))
    for name, sub in (("jdet", ""), ("jdet.models", "/models"), ("jdet.models.boxes", "/models/boxes"),
                      ("jdet.utils", "/utils"), ("jdet.data", "/data"), ("jdet.ops", "/ops"), ("jdet.config", "/config")):
        mod = types.ModuleType(name)
        mod.__path__ = [REF + sub]
        sys.modules[name] = mod
    return importlib.import_module("jdet.data.transforms"), importlib.import_module("jdet.models.boxes.box_ops")


def make_target(rng, w, h, n, with_ignore=True):
    lw = rng.uniform(12, 0.4 * min(w, h), n)
    sh = rng.uniform(4, np.minimum(lw, 0.15 * min(w, h)))
    rb = np.stack([rng.uniform(0.1 * w, 0.9 * w, n), rng.uniform(0.1 * h, 0.9 * h, n), lw, sh,
                   rng.uniform(-np.pi / 4, 3 * np.pi / 4, n)], 1).astype(np.float32)
    return rb


def main():
    if not hasattr(np, "float"):
        np.float = float      # box_ops.py:464-472 uses the alias NumPy removed in 1.24: same type
    T, B = load_reference()
    rng = np.random.default_rng(0)
    out = {}
    # ---- NumPy box helpers
    rb = make_target(rng, 800, 600, 64)
    out["boxes_in"] = rb
    out["norm_angle_in"] = rng.uniform(-7, 7, 200).astype(np.float32)
    out["norm_angle_le135"] = B.norm_angle(out["norm_angle_in"].copy(), 'le135')
    out["norm_angle_le90"] = B.norm_angle(out["norm_angle_in"].copy(), 'le90')
    for ver in ("le135", "le90"):
        polys = B.rotated_box_to_poly_np(rb.copy(), ver)
        out["r2p_" + ver] = polys
        out["p2r_" + ver] = B.poly_to_rotated_box_np(polys.copy(), ver)
    hb, pl = B.rotated_box_to_bbox_np(rb.copy())
    out["r2bbox_h"], out["r2bbox_p"] = hb, pl

    # ---- transforms on seeded images / targets
    def case(name, tf, w, h, seed, n=9):
        r = np.random.default_rng(seed)
        img = r.integers(0, 255, (h, w, 3), dtype=np.uint8)
        rboxes = make_target(r, w, h, n)
        hboxes, polys = B.rotated_box_to_bbox_np(rboxes.copy())
        tgt = dict(rboxes=rboxes.copy(), hboxes=hboxes.astype(np.float32), polys=polys.astype(np.float32),
                   labels=r.integers(1, 16, n).astype(np.int32), rboxes_ignore=make_target(r, w, h, 2),
                   img_size=(w, h), ori_img_size=(w, h), scale_factor=1.0)
        out[name + "/img_in"] = img
        for k in ("rboxes", "hboxes", "polys", "labels", "rboxes_ignore"):
            out[name + "/in_" + k] = tgt[k].copy()
        random.seed(seed)
        im2, t2 = tf(Image.fromarray(img), copy.deepcopy(tgt))
        out[name + "/img_out"] = np.asarray(im2) if isinstance(im2, Image.Image) else im2
        for k in ("rboxes", "hboxes", "polys", "rboxes_ignore"):
            out[name + "/out_" + k] = np.asarray(t2[k])
        out[name + "/meta"] = np.array([t2["img_size"][0], t2["img_size"][1], float(t2.get("scale_factor", 1.0)),
                                        *(t2.get("pad_shape", t2["img_size"])), float(t2.get("rotate_angle", -1)),
                                        {"horizontal": 1, "vertical": 2}.get(t2.get("flip"), 0)], np.float64)

    norm = dict(type="Normalize", mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_bgr=False)
    case("resize_le135", T.RotatedResize(128, 128), 100, 75, 1)
    case("resize_le90", T.RotatedResize(150, 256, angle_version='le90'), 90, 120, 2)
    case("flip_h", T.RotatedRandomFlip(prob=1.0), 80, 60, 3)
    case("flip_v", T.RotatedRandomFlip(prob=1.0, direction="vertical"), 80, 60, 4)
    # RandomRotateAug draws `int(random.random() * 100) // 25` quarter turns: seeds picked to cover 0, 1, 2, 3
    seeds, want = [], [0, 1, 2, 3]
    s = 0
    while want:
        random.seed(s)
        k = int(random.random() * 100) // 25
        if k in want:
            want.remove(k)
            seeds.append((s, k))
        s += 1
    for s, k in seeds:
        case("ra90_%d" % k, T.RandomRotateAug(random_rotate_on=True), 70, 50, s)
        assert out["ra90_%d/meta" % k][5] == 90 * k
    case("pad", T.Pad(size_divisor=32), 75, 50, 5)
    case("normalize", T.Normalize(**{k: v for k, v in norm.items() if k != "type"}), 64, 48, 6)
    case("compose_s2anet", T.Compose([dict(type="RotatedResize", min_size=128, max_size=128),
                                      dict(type="RotatedRandomFlip", prob=0.5), dict(type="Pad", size_divisor=32), norm]),
         150, 100, 7)
    case("compose_cfg4", T.Compose([dict(type="RotatedResize", min_size=128, max_size=128),
                                    dict(type="RotatedRandomFlip", prob=0.5),
                                    dict(type="RandomRotateAug", random_rotate_on=True),
                                    dict(type="Pad", size_divisor=32), norm]), 120, 120, 11)
    out["provenance"] = np.array("reference transforms.py:190-823 / box_ops.py run in the build container through "
                                 "tests/golden/make_transforms_golden.py (jittor / cv2 / skimage bound to inert "
                                 "placeholders: unused by these classes)")
    np.savez_compressed(os.path.join(HERE, "transforms.npz"), **out)
    print("wrote", len(out), "arrays;", os.path.getsize(os.path.join(HERE, "transforms.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
