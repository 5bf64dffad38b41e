"""Does MIOpen's fused convolution + bias + ReLU (torch.miopen_convolution_relu / _add_relu) run as fast as the plain
convolution on the trunk shapes?  If so, eval-mode BatchNorm folds into the weights and `conv -> bn_act` becomes one
kernel.  Prints us per call: conv2d, bn_act after it, fused conv+bias+relu, fused conv+bias+add+relu."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rs_detection_amd.utils.miopen_db import use_packaged_miopen_db  # noqa: E402
use_packaged_miopen_db()
from bench import event_time  # noqa: E402
from rs_detection_amd.ops.bn_act import bn_act  # noqa: E402

dev = torch.device("cuda")
B = 4
shapes = [  # (Cin, Cout, k, stride, H) of ResNet-50 bottlenecks at a 1024^2 tile
    (256, 64, 1, 1, 256), (64, 64, 3, 1, 256), (64, 256, 1, 1, 256),
    (512, 128, 1, 1, 128), (128, 128, 3, 1, 128), (128, 512, 1, 1, 128),
    (1024, 256, 1, 1, 64), (256, 256, 3, 1, 64), (256, 1024, 1, 1, 64),
    (2048, 512, 1, 1, 32), (512, 512, 3, 1, 32), (512, 2048, 1, 1, 32),
]
for dt in (torch.float32, torch.bfloat16):
    for cl in (True,):
        print("dtype", dt, "channels_last", cl)
        for ci, co, k, s, H in shapes:
            x = torch.randn(B, ci, H, H, device=dev, dtype=dt)
            w = torch.randn(co, ci, k, k, device=dev, dtype=dt) * 0.05
            b = torch.randn(co, device=dev, dtype=dt)
            if cl:
                x = x.contiguous(memory_format=torch.channels_last)
                w = w.contiguous(memory_format=torch.channels_last)
            bn = torch.nn.BatchNorm2d(co).to(dev).eval()
            pad = k // 2
            y = F.conv2d(x, w, None, s, pad)
            res = torch.randn_like(y)
            t_conv = event_time(lambda: F.conv2d(x, w, None, s, pad), 20, graph=False) * 1e6
            t_bn = event_time(lambda: bn_act(y, bn, None, True), 20, graph=False) * 1e6
            t_bnr = event_time(lambda: bn_act(y, bn, res, True), 20, graph=False) * 1e6
            try:
                f = lambda: torch.miopen_convolution_relu(x, w, b, [s, s], [pad, pad], [1, 1], 1)
                z = f()
                ref = torch.relu(F.conv2d(x, w, b, s, pad))
                err = float((z.float() - ref.float()).abs().max() / ref.float().abs().max())
                t_f = event_time(f, 20, graph=False) * 1e6
            except Exception as e:
                t_f, err = float("nan"), str(e)[:60]
            try:
                f2 = lambda: torch.miopen_convolution_add_relu(x, w, res, 1.0, b, [s, s], [pad, pad], [1, 1], 1)
                z2 = f2()
                t_f2 = event_time(f2, 20, graph=False) * 1e6
            except Exception as e:
                t_f2 = float("nan")
            print("  %4d->%4d k%d %3d^2: conv %7.1f  bn_act %6.1f (+res %6.1f)  fused conv+bias+relu %7.1f (err %s)  +add %7.1f"
                  % (ci, co, k, H, t_conv, t_bn, t_bnr, t_f, ("%.1e" % err) if isinstance(err, float) else err, t_f2))
