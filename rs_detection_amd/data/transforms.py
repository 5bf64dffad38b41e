"""Image / target transforms of the DOTA pipeline (/root/reference/python/jdet/data/transforms.py:190-257, :408-481,
:644-823): PIL + NumPy only (the reference uses no cv2 on this path either).  Registered in TRANSFORMS under the
reference's names so that the ``dataset.train.transforms`` section of a JDet config builds unchanged."""
import random

import numpy as np
from PIL import Image

from rs_detection_amd.utils.registry import TRANSFORMS, build_from_cfg
from .box_np import norm_angle_np, poly_to_rotated_box_np, rotated_box_to_poly_np

_BOX_KEYS = ["bboxes", "hboxes", "rboxes", "polys", "hboxes_ignore", "polys_ignore", "rboxes_ignore"]


@TRANSFORMS.register_module()
class Compose:
    """:190-207."""

    def __init__(self, transforms=None):
        self.transforms = []
        for t in (transforms or []):
            if isinstance(t, dict):
                t = build_from_cfg(t, TRANSFORMS)
            elif not callable(t):
                raise TypeError('transform must be callable or a dict')
            self.transforms.append(t)

    def __call__(self, image, target=None):
        for t in self.transforms:
            image, target = t(image, target)
        return image, target


# ---- one engine for every geometric transform of the box annotations ---------------------------------------------------
# A quarter turn, a resize and a flip all map every OUTPUT column of a box array to one simple function of ONE input
# column.  A transform is therefore a "column program" -- a list with one op per output column -- and the per-kind
# handling (horizontal boxes (n, 4), polygons (n, 8), rotated boxes (n, 5) that travel as their polygons) lives in one
# place, `_transform_boxes`.  Ops (x = the source column, arithmetic in the array's dtype, in the order written so that
# the values equal the reference's expression bit for bit):
#   ("copy", src)              x
#   ("rsub", src, c)           c - x
#   ("rsub1", src, c)          c - x - 1
#   ("scale", src, f, hi)      clip(x * f, 0, hi)
#   ("turn", src, base)        norm_angle(base - x)        (base None: norm_angle(-x))

def _run_program(boxes, program, angle_version="le135"):
    out = np.empty_like(boxes)
    for j, op in enumerate(program):
        x = boxes[..., op[1]]
        kind = op[0]
        if kind == "copy":
            col = x
        elif kind == "rsub":
            col = op[2] - x
        elif kind == "rsub1":
            col = op[2] - x - 1
        elif kind == "scale":
            col = np.clip(x * op[2], 0, op[3])
        elif kind == "turn":
            col = norm_angle_np(-x if op[2] is None else op[2] - x, angle_version)
        else:
            raise ValueError(kind)
        out[..., j] = col
    return out


def _box_kind(key):
    return "h" if ("bboxes" in key or "hboxes" in key) else ("r" if "rboxes" in key else "p")


def _transform_boxes(target, keys, program_for, angle_version="le135", rboxes_as_polys=True, need_2d=True):
    """Apply ``program_for(kind, n_columns)`` (kind 'h' / 'p' / 'r') to every box array of ``target`` named in ``keys``.
    Rotated boxes go through their corner polygons (and back) when ``rboxes_as_polys``."""
    for key in keys:
        boxes = target.get(key)
        if boxes is None or (need_2d and boxes.ndim != 2):
            continue
        kind = _box_kind(key)
        if kind == "r" and rboxes_as_polys:
            polys = rotated_box_to_poly_np(boxes, angle_version)
            polys = _run_program(polys, program_for("p", polys.shape[-1]), angle_version)
            target[key] = poly_to_rotated_box_np(polys, angle_version)
        else:
            target[key] = _run_program(boxes, program_for(kind, boxes.shape[-1]), angle_version)


@TRANSFORMS.register_module()
class RandomRotateAug:
    """:209-257: rotate by a random multiple of 90 degrees (anticlockwise), boxes with it.  One quarter turn of a
    (w, h) image sends the point (x, y) to (y, w - x)."""

    def __init__(self, angle_version='le135', random_rotate_on=False):
        self.random_rotate_on, self.angle_version = random_rotate_on, angle_version

    @staticmethod
    def _quarter_turn(kind, ncols, w):
        if kind == "h":      # (x0, y0, x1, y1) -> (y0, w - x1, y1, w - x0): corners 0 / 1 trade their new-y roles
            return [("copy", 1), ("rsub", 2, w), ("copy", 3), ("rsub", 0, w)]
        prog = []
        for k in range(ncols // 2):
            prog += [("copy", 2 * k + 1), ("rsub", 2 * k, w)]
        return prog

    def _rotate_boxes_90(self, target, size):
        w = size[0]
        _transform_boxes(target, _BOX_KEYS, lambda kind, n: self._quarter_turn(kind, n, w), self.angle_version)

    def __call__(self, image, target=None):
        if self.random_rotate_on:
            turns = int(random.random() * 100) // 25
            for _ in range(turns):
                if target is not None:
                    self._rotate_boxes_90(target, image.size)
                image = image.rotate(90, expand=True)
            if target is not None:
                target["rotate_angle"] = 90 * turns
        return image, target


@TRANSFORMS.register_module()
class Resize:
    """:408-481: the short side goes to a size drawn from ``min_size`` (kept within 1.5 x of the original and so that
    the long side stays under ``max_size``), the long side follows; ``keep_ratio=False``: (min_size[0], max_size)."""
    _keys = ("bboxes", "polys")

    def __init__(self, min_size, max_size, keep_ratio=True):
        self.min_size = tuple(min_size) if isinstance(min_size, (list, tuple)) else (min_size,)
        self.max_size, self.keep_ratio = max_size, keep_ratio

    def get_size(self, image_size):
        w, h = image_size
        size = random.choice(self.min_size)            # drawn in either mode (the random stream of the reference)
        if not self.keep_ratio:
            oh, ow = self.min_size[0], self.max_size
            return (int(oh), int(ow)), oh / h
        short, long_ = (w, h) if w <= h else (h, w)
        size = np.clip(size, int(short / 1.5), int(short * 1.5))
        if self.max_size is not None and float(long_) / float(short) * size > self.max_size:
            size = int(round(self.max_size * float(short) / float(long_)))
        if short == size:
            return (h, w), 1.
        other = int(size * long_ / short)
        oh, ow = (other, size) if w < h else (size, other)
        assert np.abs(oh / h - ow / w) < 1e-2
        return (int(oh), int(ow)), oh / h

    def _resize_boxes(self, target, size):
        width, height = target["img_size"]
        new_w, new_h = size
        fx, fy = float(new_w / width), float(new_h / height)

        def program(kind, ncols):
            return [("scale", c, fy, new_h - 1) if c % 2 else ("scale", c, fx, new_w - 1) for c in range(ncols)]
        _transform_boxes(target, self._keys, program, getattr(self, "angle_version", "le135"))

    def __call__(self, image, target=None):
        size, scale_factor = self.get_size(image.size)
        image = image.resize(size[::-1], Image.BILINEAR)
        if target is not None:
            self._resize_boxes(target, image.size)
            target["img_size"] = image.size
            target["scale_factor"] = scale_factor
            target["pad_shape"] = image.size
            target["keep_ratio"] = self.keep_ratio
        return image, target


@TRANSFORMS.register_module()
class RotatedResize(Resize):
    """:644-678: every box kind; rotated boxes go through their polygons (scale, clip to the border, back to a box)."""
    _keys = tuple(_BOX_KEYS)

    def __init__(self, min_size, max_size, angle_version='le135', keep_ratio=True):
        super().__init__(min_size, max_size, keep_ratio)
        self.angle_version = angle_version


@TRANSFORMS.register_module()
class RandomFlip:
    """:680-723.  Its box rule is the horizontal-box one applied to whatever array it meets (columns in groups of four:
    x0, y0, x1, y1), polygons included -- the reference's behaviour, kept."""
    _keys = ("bboxes", "polys")

    def __init__(self, prob=0.5, direction="horizontal"):
        assert direction in ['horizontal', 'vertical', 'diagonal'], f"{direction} not supported"
        self.direction, self.prob = direction, prob

    def _hbox_program(self, ncols, w, h):
        flip_x = self.direction in ('horizontal', 'diagonal')
        flip_y = self.direction in ('vertical', 'diagonal')
        prog = []
        for c in range(ncols):
            r = c % 4
            partner = c + 2 if r < 2 else c - 2          # x0 <-> x1, y0 <-> y1 of the same group of four
            on = flip_x if r % 2 == 0 else flip_y
            if on and partner < ncols:
                prog.append(("rsub", partner, w if r % 2 == 0 else h))       # x0' = w - x1, x1' = w - x0 (y alike)
            else:
                prog.append(("copy", c))
        return prog

    def _program(self, kind, ncols, w, h):
        return self._hbox_program(ncols, w, h)

    def _flip_boxes(self, target, size):
        w, h = self._extent(target, size)
        _transform_boxes(target, self._keys, lambda kind, n: self._program(kind, n, w, h),
                         rboxes_as_polys=False, need_2d=False)

    def _extent(self, target, size):
        return target["img_size"]

    def _flip_image(self, image):
        if self.direction in ("horizontal", "diagonal"):
            image = image.transpose(Image.FLIP_LEFT_RIGHT)
        if self.direction in ("vertical", "diagonal"):
            image = image.transpose(Image.FLIP_TOP_BOTTOM)
        return image

    def __call__(self, image, target=None):
        if random.random() < self.prob:
            image = self._flip_image(image)
            if target is not None:
                self._flip_boxes(target, image.size)
                target["flip"] = self.direction
        return image, target


@TRANSFORMS.register_module()
class RotatedRandomFlip(RandomFlip):
    """:725-777: points mirror as x -> w - x - 1; a rotated box keeps its size, mirrors its centre and turns its angle
    (horizontal: pi - a, vertical: -a; no diagonal flip for rotated boxes, :736-739)."""
    _keys = tuple(_BOX_KEYS)

    def _extent(self, target, size):
        return size

    def _program(self, kind, ncols, w, h):
        if kind == "h":
            return self._hbox_program(ncols, w, h)
        flip_x = self.direction in ('horizontal', 'diagonal')
        flip_y = self.direction in ('vertical', 'diagonal')
        if kind == "p":
            return [(("rsub1", c, h) if flip_y else ("copy", c)) if c % 2 else (("rsub1", c, w) if flip_x else ("copy", c))
                    for c in range(ncols)]
        assert self.direction in ('horizontal', 'vertical'), "rotated boxes: horizontal / vertical flips only (:736-739)"
        prog = []
        for c in range(ncols):
            r = c % 5
            if r == 0 and flip_x:
                prog.append(("rsub1", c, w))
            elif r == 1 and flip_y:
                prog.append(("rsub1", c, h))
            elif r == 4:
                prog.append(("turn", c, np.pi if flip_x else None))
            else:
                prog.append(("copy", c))
        return prog


@TRANSFORMS.register_module()
class Pad:
    """:779-800."""

    def __init__(self, size=None, size_divisor=None, pad_val=0):
        assert (size is None) != (size_divisor is None)
        self.size, self.size_divisor, self.pad_val = size, size_divisor, pad_val

    def __call__(self, image, target=None):
        if self.size is not None:
            pad_w, pad_h = self.size
        else:
            pad_h = int(np.ceil(image.size[1] / self.size_divisor)) * self.size_divisor
            pad_w = int(np.ceil(image.size[0] / self.size_divisor)) * self.size_divisor
        new_image = Image.new(image.mode, (pad_w, pad_h), (self.pad_val,) * len(image.split()))
        new_image.paste(image, (0, 0, image.size[0], image.size[1]))
        if target is not None:
            target["pad_shape"] = new_image.size
        return new_image, target


@TRANSFORMS.register_module()
class Normalize:
    """:803-823."""

    def __init__(self, mean, std, to_bgr=True):
        self.mean = np.float32(mean).reshape(-1, 1, 1)
        self.std = np.float32(std).reshape(-1, 1, 1)
        self.to_bgr = to_bgr

    def __call__(self, image, target=None):
        if isinstance(image, Image.Image):
            image = np.array(image).transpose((2, 0, 1))
        if self.to_bgr:
            image = image[::-1]
        image = (image - self.mean) / self.std
        if target is not None:
            target["mean"], target["std"], target["to_bgr"] = self.mean, self.std, self.to_bgr
        return image, target
