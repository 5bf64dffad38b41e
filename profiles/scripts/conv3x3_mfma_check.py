"""rsdet_conv3x3_fwd_mfma_bf16 against F.conv2d (values) and MIOpen / CK (time) on the head-canvas shape and a few odd
ones.  Usage: python profiles/scripts/conv3x3_mfma_check.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rs_detection_amd.utils.miopen_db import use_packaged_miopen_db  # noqa: E402
use_packaged_miopen_db()
from bench import event_time  # noqa: E402
from rs_detection_amd import _lib  # noqa: E402

dev = torch.device("cuda")
lib = _lib.load()


def ours(x, w, bias=None, live=None, relu=False):
    B, C, H, W = x.shape
    O = w.shape[0]
    out = torch.empty((B, O, H, W), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
    rc = lib.rsdet_conv3x3_fwd_mfma_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(live), B, H, W, C, O, int(relu),
                                         _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "rsdet_conv3x3_fwd_mfma_bf16")
    return out


torch.manual_seed(0)
for (B, C, O, H, W) in [(4, 256, 256, 128, 196), (1, 64, 32, 5, 37), (2, 128, 96, 9, 300), (1, 256, 256, 3, 224), (2, 64, 256, 17, 1)]:
    x = torch.randn(B, C, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(O, C, 3, 3, device=dev) * 0.05).bfloat16().contiguous(memory_format=torch.channels_last)
    bias = torch.randn(O, device=dev)
    live = (torch.rand(H, W, device=dev) > 0.2).to(torch.uint8).contiguous()       # one mask for all images
    ref = F.conv2d(x.float(), w.float(), None, 1, 1)
    got = ours(x, w)
    e0 = float((got.float() - ref).abs().max() / ref.abs().max())
    lib_y = F.conv2d(x, w, None, 1, 1)
    e_lib = float((lib_y.float() - ref).abs().max() / ref.abs().max())
    ref2 = torch.relu(ref + bias[None, :, None, None]) * live[None, None].float()
    got2 = ours(x, w, bias, live, True)
    e1 = float((got2.float() - ref2).abs().max() / ref2.abs().max())
    assert got.is_contiguous(memory_format=torch.channels_last)
    print("B%d C%d O%d %dx%d: rel err vs fp32 conv %.2e (MIOpen bf16: %.2e); with bias+relu+live %.2e" % (B, C, O, H, W, e0, e_lib, e1))
    if (B, C, O, H, W) == (4, 256, 256, 128, 196):
        fl = 2.0 * B * H * W * O * 9 * C
        t = sorted(event_time(lambda: ours(x, w), 20) for _ in range(3))[1]
        t2 = sorted(event_time(lambda: ours(x, w, bias, live, True), 20) for _ in range(3))[1]
        tl = sorted(event_time(lambda: F.conv2d(x, w, None, 1, 1), 20) for _ in range(3))[1]
        print("  canvas shape: ours %.1f us = %.0f TFLOP/s (%.3f of 2.5 PF); fused epilogue %.1f us; MIOpen/CK %.1f us = %.0f TFLOP/s"
              % (t * 1e6, fl / t / 1e12, fl / t / 2.5e15, t2 * 1e6, tl * 1e6, fl / tl / 1e12))
