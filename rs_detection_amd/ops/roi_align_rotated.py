"""jdet.ops.roi_align_rotated on MI355X: ROIAlignRotated (the detectron2-style variant).

Mirror of /root/reference/python/jdet/ops/roi_align_rotated.py:256-330.  Same kernels as ROIAlignRotated_v1
(csrc/rroi_align.hip); the RoI frame has no -0.5 pixel shift (:76-77) and the opposite rotation sense (:116-117).
Default sampling_ratio = 0 (adaptive grid, :90-94), as in the reference module (:312).
"""
import torch.nn as nn

from .roi_align_rotated_v1 import _pair, rroi_align

__all__ = ["ROIAlignRotated", "roi_align"]


def roi_align(input, rois, output_size, spatial_scale, sampling_ratio):
    """_RotatedROIAlign.apply (:309): input (N,C,H,W), rois (R,6) = (batch, cx, cy, w, h, theta[rad])."""
    return rroi_align(input, rois, _pair(output_size), spatial_scale, sampling_ratio, "v0")


class ROIAlignRotated(nn.Module):
    def __init__(self, output_size, spatial_scale, sampling_ratio=0):
        super().__init__()
        self.output_size = _pair(output_size)
        self.spatial_scale = spatial_scale
        self.sampling_ratio = sampling_ratio

    def forward(self, input, rois):
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio)

    def __repr__(self):
        return "%s(output_size=%s, spatial_scale=%s, sampling_ratio=%s)" % (
            self.__class__.__name__, self.output_size, self.spatial_scale, self.sampling_ratio)
