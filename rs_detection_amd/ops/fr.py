"""jdet.ops.fr on MI355X: FeatureRefine (R3Det).

Mirror of /root/reference/python/jdet/ops/fr.py:234-347 (feature_refine, FR, FeatureRefineModule); kernels in
csrc/feature_refine.hip.  No CPU fallback.
"""
import torch
import torch.nn as nn

from .. import _lib
from .layout import nchw_to_nhwc, nhwc_to_nchw

__all__ = ["feature_refine", "FR", "FeatureRefineModule"]


class FeatureRefineFunction(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, features, best_rbboxes, spatial_scale, points=1):
        assert points in [1, 5]  # fr.py:261
        _lib.require_cuda_f32(features, best_rbboxes)
        lib = _lib.load()
        N, C, H, W = features.shape
        # a channels_last map stays channels_last both ways (csrc/feature_refine.hip: the NHWC forward; the backward's
        # gather is channels-last natively): no layout turn in either direction
        cl = (features.dim() == 4 and not features.is_contiguous()
              and features.is_contiguous(memory_format=torch.channels_last)
              and bool(lib.rsdet_feature_refine_forward_nhwc_supported(C)))
        best_rbboxes = best_rbboxes.contiguous()
        if not cl:
            features = features.contiguous()
        if best_rbboxes.numel() != N * H * W * 5:
            raise _lib.RsdetError("best_rbboxes must hold (N, H, W, 5) = %s boxes, got %s"
                                  % ((N, H, W, 5), tuple(best_rbboxes.shape)))
        ctx.save_for_backward(best_rbboxes)
        ctx.cfg = (float(spatial_scale), int(points))
        out = torch.empty_like(features)
        if cl:
            rc = lib.rsdet_feature_refine_forward_nhwc_f32(_lib.ptr(features), _lib.ptr(best_rbboxes), N, C, H, W,
                                                           float(spatial_scale), int(points), _lib.ptr(out),
                                                           _lib.stream_ptr())
            _lib.check(rc, "rsdet_feature_refine_forward_nhwc_f32")
            return out
        rc = lib.rsdet_feature_refine_forward_f32(_lib.ptr(features), _lib.ptr(best_rbboxes), N, C, H, W,
                                                  float(spatial_scale), int(points), _lib.ptr(out), _lib.stream_ptr())
        _lib.check(rc, "rsdet_feature_refine_forward_f32")
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, grad_output):
        (boxes,) = ctx.saved_tensors
        scale, points = ctx.cfg
        return feature_refine_backward(grad_output, boxes, scale, points), None, None, None


def feature_refine_backward(grad_output, boxes, scale, points):
    """grad_features of FeatureRefineFunction (fr.py:235-260); a plain function so that the bench can replay it from a
    hipGraph (device time, like the forward rows)."""
    lib = _lib.load()
    N, C, H, W = grad_output.shape
    if N == 0 or C == 0:
        return torch.zeros_like(grad_output)
    cl = not grad_output.is_contiguous() and grad_output.is_contiguous(memory_format=torch.channels_last)
    # channels-last rows for the gather: a channels_last gradient IS that matrix, an NCHW one is turned (and turned back)
    go = grad_output.permute(0, 2, 3, 1) if cl else nchw_to_nhwc(grad_output.contiguous())
    gi = torch.empty_like(go)
    ws_bytes = lib.rsdet_feature_refine_backward_ws_size(N, H, W, points)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=go.device)
    rc = lib.rsdet_feature_refine_backward_nhwc_f32(_lib.ptr(go), _lib.ptr(boxes), N, C, H, W, scale, points,
                                                    _lib.ptr(gi), _lib.ptr(ws), ws_bytes, _lib.stream_ptr())
    _lib.check(rc, "rsdet_feature_refine_backward_nhwc_f32")
    return gi.permute(0, 3, 1, 2) if cl else nhwc_to_nchw(gi)


def feature_refine(features, best_rbboxes, spatial_scale, points=1):
    return FeatureRefineFunction.apply(features, best_rbboxes, spatial_scale, points)


class FR(nn.Module):
    """fr.py:275-288."""

    def __init__(self, spatial_scale, points=1):
        super().__init__()
        self.spatial_scale = float(spatial_scale)
        self.points = points

    def forward(self, features, best_rbboxes):
        return feature_refine(features, best_rbboxes, self.spatial_scale, self.points)

    def __repr__(self):
        return "%s(spatial_scale=%s, points=%s)" % (self.__class__.__name__, self.spatial_scale, self.points)


class FeatureRefineModule(nn.Module):
    """fr.py:291-347: (5x1 o 1x5) + 1x1 convolutions, FR at every level, residual add.  Parameter names match the
    reference (`conv_5_1`, `conv_1_5`, `conv_1_1`) for checkpoint import."""

    def __init__(self, in_channels, featmap_strides, conv_cfg=None, norm_cfg=None):
        super().__init__()
        self.in_channels = in_channels
        self.featmap_strides = featmap_strides
        self.conv_cfg = conv_cfg
        self.norm_cfg = norm_cfg
        self.fr = nn.ModuleList([FR(spatial_scale=1 / s) for s in self.featmap_strides])
        self.conv_5_1 = nn.Conv2d(in_channels, in_channels, kernel_size=(5, 1), stride=1, padding=(2, 0))
        self.conv_1_5 = nn.Conv2d(in_channels, in_channels, kernel_size=(1, 5), stride=1, padding=(0, 2))
        self.conv_1_1 = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.init_weights()

    def init_weights(self):
        for m in (self.conv_5_1, self.conv_1_5, self.conv_1_1):  # normal_init(std=0.01), bias 0 (:326-329)
            nn.init.normal_(m.weight, mean=0.0, std=0.01)
            nn.init.constant_(m.bias, 0.0)

    def forward(self, x, best_rbboxes):
        """x: list of per-level feature maps; best_rbboxes: per image, per level (H*W, 5) best boxes (:331-346)."""
        mlvl_rbboxes = [torch.cat(best_rbbox) for best_rbbox in zip(*best_rbboxes)]
        out = []
        for x_scale, best_rbboxes_scale, fr_scale in zip(x, mlvl_rbboxes, self.fr):
            feat_scale = self.conv_5_1(self.conv_1_5(x_scale)) + self.conv_1_1(x_scale)
            out.append(x_scale + fr_scale(feat_scale, best_rbboxes_scale))
        return out
