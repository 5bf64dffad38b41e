"""rsdet_conv3x3_wrw_mfma_bf16 against the fp32 weight gradient (values) and MIOpen (time) on the head-canvas shape and
a few odd ones.  Usage: python profiles/scripts/conv3x3_wrw_check.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rs_detection_amd.utils.miopen_db import use_packaged_miopen_db  # noqa: E402
use_packaged_miopen_db()
from bench import event_time  # noqa: E402
from rs_detection_amd import _lib  # noqa: E402

dev = torch.device("cuda")
lib = _lib.load()


def ours(g, x, out_bf16=True):
    B, C, H, W = x.shape
    O = g.shape[1]
    gw = torch.empty((O, C, 3, 3), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=dev,
                     memory_format=torch.channels_last)
    nb = lib.rsdet_conv3x3_wrw_mfma_ws_size(B, H, W, C, O)
    ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
    rc = lib.rsdet_conv3x3_wrw_mfma_bf16(_lib.ptr(g), _lib.ptr(x), B, H, W, C, O, _lib.ptr(gw), int(out_bf16), _lib.ptr(ws),
                                         nb, _lib.stream_ptr())
    _lib.check(rc, "rsdet_conv3x3_wrw_mfma_bf16")
    return gw


def lib_wrw(g, x, w):
    return torch.ops.aten.convolution_backward(g, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False))[1]


torch.manual_seed(0)
for (B, C, O, H, W) in [(4, 256, 256, 128, 196), (1, 64, 32, 5, 37), (2, 128, 96, 9, 300), (1, 256, 256, 3, 224), (2, 64, 256, 17, 1),
                        (3, 64, 8, 2, 33)]:
    x = torch.randn(B, C, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, O, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.zeros(O, C, 3, 3, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    ref = lib_wrw(g.float().contiguous(), x.float().contiguous(), w.float().contiguous())
    got = ours(g, x, False)
    e0 = float((got - ref).abs().max() / ref.abs().max())
    got16 = ours(g, x, True)
    e1 = float((got16.float() - ref).abs().max() / ref.abs().max())
    lw = lib_wrw(g, x, w)
    e_lib = float((lw.float() - ref).abs().max() / ref.abs().max())
    print("B%d C%d O%d %dx%d: rel err vs fp32 wrw: fp32 out %.2e, bf16 out %.2e (MIOpen bf16: %.2e)" % (B, C, O, H, W, e0, e1, e_lib))
    if (B, C, O, H, W) == (4, 256, 256, 128, 196):
        fl = 2.0 * B * H * W * O * 9 * C
        t = sorted(event_time(lambda: ours(g, x), 20, graph=False) for _ in range(3))[1]
        tl = sorted(event_time(lambda: lib_wrw(g, x, w), 20, graph=False) for _ in range(3))[1]
        print("  canvas shape: ours %.1f us = %.0f TFLOP/s (%.3f of 2.5 PF); MIOpen %.1f us = %.0f TFLOP/s"
              % (t * 1e6, fl / t / 1e12, fl / t / 2.5e15, tl * 1e6, fl / tl / 1e12))
