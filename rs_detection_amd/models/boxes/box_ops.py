"""Box conversions on the S2ANet path (reference: /root/reference/python/jdet/models/boxes/box_ops.py).
The arithmetic lives in csrc/box_coder.hip; these are the reference-named entry points."""
import numpy as np
import torch

from rs_detection_amd.ops.box_coder import (bbox2delta_rotated, delta2bbox_rotated, rotated_box_to_poly)  # noqa: F401


def norm_angle(angle, angle_version='le135'):
    """box_ops.py:176-182 (torch remainder == Python-style mod)."""
    lo = float(-np.pi / 2) if angle_version == 'le90' else float(-np.pi / 4)
    return torch.remainder(angle - lo, float(np.pi)) + lo
