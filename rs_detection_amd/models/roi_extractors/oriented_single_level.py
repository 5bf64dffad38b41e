"""OrientedSingleRoIExtractor (/root/reference/python/jdet/models/roi_extractors/oriented_single_level.py:9-114).

Level = floor(log2(sqrt(w*h)/56 + 1e-6)) clamped, computed AFTER the (1.4, 1.2) extension (the first mapping at
:97 is overwritten at :102); ``roi_rescale`` multiplies column 3 (w) by the SECOND factor and column 4
(h) by the first (:85-88).  RROIAlign per level: csrc/rroi_align.hip."""
import torch
import torch.nn as nn

import importlib

_rr = importlib.import_module("rs_detection_amd.ops.roi_align_rotated_v1")  # the module (ops/__init__ re-exports a function of the same name)
from rs_detection_amd.utils.registry import ROI_EXTRACTORS


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


@ROI_EXTRACTORS.register_module()
class OrientedSingleRoIExtractor(nn.Module):
    def __init__(self, roi_layer, out_channels, featmap_strides, extend_factor=(1., 1.), finest_scale=56):
        super().__init__()
        self.roi_layers = self.build_roi_layers(roi_layer, featmap_strides)
        self.out_channels, self.featmap_strides = out_channels, featmap_strides
        self.extend_factor, self.finest_scale = extend_factor, finest_scale
        # the fused forward serves layers that differ in their scale only (RSDET_RROI_LEVELS=0: the per-level form)
        import os
        from rs_detection_amd.ops.roi_align_rotated import ROIAlignRotated as _V0
        same = all(type(l) is type(self.roi_layers[0]) and l.output_size == self.roi_layers[0].output_size
                   and l.sampling_ratio == self.roi_layers[0].sampling_ratio for l in self.roi_layers)
        kind = type(self.roi_layers[0])
        self._levels_variant = ("v1" if kind is _rr.ROIAlignRotated_v1 else "v0" if kind is _V0 else None) if same else None
        self._one_launch = os.environ.get("RSDET_RROI_LEVELS", "1") != "0"

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def init_weights(self):
        pass

    def build_roi_layers(self, layer_cfg, featmap_strides):
        cfg = dict(layer_cfg)
        layer_type = cfg.pop('type')
        assert hasattr(_rr, layer_type)
        cls = getattr(_rr, layer_type)
        return nn.ModuleList([cls(spatial_scale=1 / s, **cfg) for s in featmap_strides])

    def map_roi_levels(self, rois, num_levels):
        scale = torch.sqrt(rois[:, 3] * rois[:, 4])
        lvls = torch.floor(torch.log2(scale / self.finest_scale + 1e-6))
        return lvls.clamp(min=0, max=num_levels - 1).long()

    def roi_rescale(self, rois, scale_factor):
        if scale_factor is None:
            return rois
        h_f, w_f = _pair(scale_factor)
        new = rois.clone()
        new[:, 3] = w_f * new[:, 3]
        new[:, 4] = h_f * new[:, 4]
        return new

    def forward(self, feats, rois, roi_scale_factor=None):
        if len(feats) == 1:
            return self.roi_layers[0](feats[0], rois)
        out_size = self.roi_layers[0].output_size[0]
        num_levels = len(feats)
        rois = self.roi_rescale(rois, self.extend_factor)
        target_lvls = self.map_roi_levels(rois, num_levels)
        rois = self.roi_rescale(rois, roi_scale_factor)
        if self._one_launch and self._levels_variant is not None and _rr.rroi_align_levels_applies(feats, rois):
            # one launch: every RoI samples the map of its own level (the per-level form below: four launches over all
            # RoIs + three adds of the (R, C, 7, 7) result)
            return _rr.rroi_align_levels(feats, rois, target_lvls, self.roi_layers[0].output_size,
                                         [l.spatial_scale for l in self.roi_layers[:num_levels]],
                                         self.roi_layers[0].sampling_ratio, self._levels_variant)
        # No per-level boolean gather / scatter and no host `any()`: every level aligns ALL rois with the
        # wrong-level rois pushed far outside the map (their samples read as 0, ROIAlignRotatedForward :30-32)
        # -- cheaper than the data-dependent index sets and sync-free.
        roi_feats = rois.new_zeros((rois.shape[0], self.out_channels, out_size, out_size))
        for i in range(num_levels):
            on = (target_lvls == i)
            r = rois.clone()
            r[:, 1:3] = torch.where(on[:, None], rois[:, 1:3], -1e8)
            r[:, 3:5] = torch.where(on[:, None], rois[:, 3:5], 1.0)
            roi_feats = roi_feats + self.roi_layers[i](feats[i], r)
        return roi_feats
