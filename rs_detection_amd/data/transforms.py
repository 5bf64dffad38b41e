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


@TRANSFORMS.register_module()
class RandomRotateAug:
    """:209-257: rotate by a random multiple of 90 degrees (anticlockwise), boxes with it."""

    def __init__(self, angle_version='le135', random_rotate_on=False):
        self.random_rotate_on, self.angle_version = random_rotate_on, angle_version

    def _rotate_boxes_90(self, target, size):
        w, h = size
        for key in _BOX_KEYS:
            if key not in target:
                continue
            bboxes = target[key]
            if bboxes.ndim < 2:
                continue
            if "bboxes" in key or "hboxes" in key:
                new_boxes = np.zeros_like(bboxes)
                new_boxes[:, ::2] = bboxes[:, 1::2]       # x = y
                new_boxes[:, 1] = w - bboxes[:, 2]        # y = w - x
                new_boxes[:, 3] = w - bboxes[:, 0]
                target[key] = new_boxes
                continue
            if "rboxes" in key:
                bboxes = rotated_box_to_poly_np(bboxes, self.angle_version)
            new_bboxes = np.zeros_like(bboxes)
            new_bboxes[:, 0::2] = bboxes[:, 1::2]
            new_bboxes[:, 1::2] = w - bboxes[:, 0::2]
            if "rboxes" in key:
                new_bboxes = poly_to_rotated_box_np(new_bboxes, self.angle_version)
            target[key] = new_bboxes

    def __call__(self, image, target=None):
        if self.random_rotate_on:
            indx = int(random.random() * 100) // 25
            for _ in range(indx):
                if target is not None:
                    self._rotate_boxes_90(target, image.size)
                image = image.rotate(90, expand=True)
            if target is not None:
                target["rotate_angle"] = 90 * indx
        return image, target


@TRANSFORMS.register_module()
class Resize:
    """:408-481."""

    def __init__(self, min_size, max_size, keep_ratio=True):
        self.min_size = tuple(min_size) if isinstance(min_size, (list, tuple)) else (min_size,)
        self.max_size, self.keep_ratio = max_size, keep_ratio

    def get_size(self, image_size):
        w, h = image_size
        size = random.choice(self.min_size)
        max_size = self.max_size
        if self.keep_ratio:
            size = np.clip(size, int(w / 1.5), int(w * 1.5)) if w <= h else np.clip(size, int(h / 1.5), int(h * 1.5))
            if max_size is not None:
                mn, mx = float(min((w, h))), float(max((w, h)))
                if mx / mn * size > max_size:
                    size = int(round(max_size * mn / mx))
            if (w <= h and w == size) or (h <= w and h == size):
                return (h, w), 1.
            if w < h:
                ow, oh = size, int(size * h / w)
            else:
                oh, ow = size, int(size * w / h)
            assert np.abs(oh / h - ow / w) < 1e-2
        else:
            oh, ow = self.min_size[0], self.max_size
        return (int(oh), int(ow)), oh / h

    def _scale_clip(self, bboxes, target, size):
        width, height = target["img_size"]
        new_w, new_h = size
        bboxes[:, 0::2] = bboxes[:, 0::2] * float(new_w / width)
        bboxes[:, 1::2] = bboxes[:, 1::2] * float(new_h / height)
        bboxes[:, 0::2] = np.clip(bboxes[:, 0::2], 0, new_w - 1)
        bboxes[:, 1::2] = np.clip(bboxes[:, 1::2], 0, new_h - 1)
        return bboxes

    def _resize_boxes(self, target, size):
        for key in ["bboxes", "polys"]:
            if key in target:
                target[key] = self._scale_clip(target[key], target, size)

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
    """:644-678: rotated boxes go through their polygons (scale, clip to the border, back to a rotated box)."""

    def __init__(self, min_size, max_size, angle_version='le135', keep_ratio=True):
        super().__init__(min_size, max_size, keep_ratio)
        self.angle_version = angle_version

    def _resize_boxes(self, target, size):
        for key in _BOX_KEYS:
            if key not in target:
                continue
            bboxes = target[key]
            if bboxes is None or bboxes.ndim != 2:
                continue
            if "rboxes" in key:
                bboxes = rotated_box_to_poly_np(bboxes, self.angle_version)
            bboxes = self._scale_clip(bboxes, target, size)
            if "rboxes" in key:
                bboxes = poly_to_rotated_box_np(bboxes, self.angle_version)
            target[key] = bboxes


@TRANSFORMS.register_module()
class RandomFlip:
    """:680-723."""

    def __init__(self, prob=0.5, direction="horizontal"):
        assert direction in ['horizontal', 'vertical', 'diagonal'], f"{direction} not supported"
        self.direction, self.prob = direction, prob

    def _flip_hboxes(self, bboxes, w, h):
        flipped = bboxes.copy()
        if self.direction in ('horizontal', 'diagonal'):
            flipped[..., 0::4] = w - bboxes[..., 2::4]
            flipped[..., 2::4] = w - bboxes[..., 0::4]
        if self.direction in ('vertical', 'diagonal'):
            flipped[..., 1::4] = h - bboxes[..., 3::4]
            flipped[..., 3::4] = h - bboxes[..., 1::4]
        return flipped

    def _flip_boxes(self, target, size):
        w, h = target["img_size"]
        for key in ["bboxes", "polys"]:
            if key in target:
                target[key] = self._flip_hboxes(target[key], w, h)

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
    """:725-777."""

    def _flip_rboxes(self, bboxes, w, h):
        flipped = bboxes.copy()
        if self.direction == 'horizontal':
            flipped[..., 0::5] = w - flipped[..., 0::5] - 1
            flipped[..., 4::5] = norm_angle_np(np.pi - flipped[..., 4::5])
        elif self.direction == 'vertical':
            flipped[..., 1::5] = h - flipped[..., 1::5] - 1
            flipped[..., 4::5] = norm_angle_np(-flipped[..., 4::5])
        else:
            assert False, "rotated boxes: horizontal / vertical flips only (:736-739)"
        return flipped

    def _flip_polys(self, bboxes, w, h):
        flipped = bboxes.copy()
        if self.direction in ('horizontal', 'diagonal'):
            flipped[..., 0::2] = w - flipped[..., 0::2] - 1
        if self.direction in ('vertical', 'diagonal'):
            flipped[..., 1::2] = h - flipped[..., 1::2] - 1
        return flipped

    def _flip_boxes(self, target, size):
        w, h = size
        for key in _BOX_KEYS:
            if key not in target:
                continue
            bboxes = target[key]
            if "rboxes" in key:
                target[key] = self._flip_rboxes(bboxes, w, h)
            elif "polys" in key:
                target[key] = self._flip_polys(bboxes, w, h)
            else:
                target[key] = self._flip_hboxes(bboxes, w, h)


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
