"""NCHW <-> NHWC turns of fp32 tensors through csrc/layout.hip (64 x 64 LDS tiles, 16-byte accesses on both sides):
the channels-last gather kernels of RROIAlign / FeatureRefine / AlignConv consume and produce channels-last, their
callers (roi_align_rotated_v1.py:329-351, fr.py:235-260, dcn_v1.py:456-557 of the reference) NCHW.  torch's generic
strided copy runs such a turn at ~1.5 TB/s; anything the kernel does not cover (other dtypes, CPU) takes torch's."""
import torch

from .. import _lib


def transpose_last2(x):
    """(B, R, C) contiguous -> (B, C, R) contiguous."""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.is_contiguous() and x.shape[1] <= 64 * 65535 \
            and x.shape[0] <= 65535:
        B, R, C = x.shape
        out = torch.empty((B, C, R), dtype=x.dtype, device=x.device)
        _lib.check(_lib.load().rsdet_transpose_last2_f32(_lib.ptr(x), _lib.ptr(out), B, R, C, _lib.stream_ptr()),
                   "rsdet_transpose_last2_f32")
        return out
    return x.transpose(1, 2).contiguous()


def nchw_to_nhwc(x):
    """(N, C, H, W) -> a contiguous (N, H, W, C) tensor."""
    N, C, H, W = x.shape
    if x.is_contiguous():
        return transpose_last2(x.view(N, C, H * W)).view(N, H, W, C)
    return x.permute(0, 2, 3, 1).contiguous()


def nhwc_to_nchw(x):
    """contiguous (N, H, W, C) -> a contiguous (N, C, H, W) tensor."""
    N, H, W, C = x.shape
    if x.is_contiguous():
        return transpose_last2(x.view(N, H * W, C)).view(N, C, H, W)
    return x.permute(0, 3, 1, 2).contiguous()
