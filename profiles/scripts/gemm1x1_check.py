"""csrc/gemm1x1_mfma.hip: values against the fp32 reference (conv -> eval BatchNorm -> + identity -> relu) and device
time against the two-launch form (library GEMM + bn_act pass) on the ResNet-50 trunk shapes of the 4 x 1024^2 bf16 step."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import event_time  # noqa: E402
from rs_detection_amd.ops import conv_bn  # noqa: E402
from rs_detection_amd.ops.bn_act import bn_act  # noqa: E402
from rs_detection_amd.ops.conv1x1 import conv1x1  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [(4, 256, 128, 256, False), (4, 512, 128, 128, False), (4, 128, 512, 128, True), (4, 512, 256, 128, False),
          (4, 1024, 256, 64, False), (4, 256, 1024, 64, True), (4, 1024, 512, 64, False), (4, 2048, 512, 32, False),
          (4, 512, 2048, 32, True), (4, 64, 64, 256, False), (4, 64, 256, 256, True), (2, 128, 96, 37, True)]
tot_f = tot_t = 0.0
for (B, C, O, H, res) in shapes:
    x = torch.randn(B, C, H, H, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(C, O, 1, bias=False).to(dev)
    conv.weight.data = (torch.randn(O, C, 1, 1, device=dev) / C ** 0.5).bfloat16()
    bn = torch.nn.BatchNorm2d(O).to(dev).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5), bn.bias.normal_(0, 0.3), bn.running_mean.normal_(0, 0.3), bn.running_var.uniform_(0.5, 2)
    r = torch.randn(B, O, H, H, device=dev).bfloat16().contiguous(memory_format=torch.channels_last) if res else None
    ok = conv_bn.conv_bn_act_applies(conv, bn, x, r)
    ref = F.batch_norm(F.conv2d(x.float(), conv.weight.float()), bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
    ref = torch.relu(ref + (r.float() if res else 0))
    y = conv_bn.conv_bn_act(conv, bn, x, residual=r)
    err = float((y.float() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    conv_bn._ON = False
    y2 = conv_bn.conv_bn_act(conv, bn, x, residual=r)
    conv_bn._ON = True
    err2 = float((y2.float() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    tf = event_time(lambda: conv_bn.conv_bn_act(conv, bn, x, residual=r), 20, 3) * 1e6

    def two():
        conv_bn._ON = False
        out = conv_bn.conv_bn_act(conv, bn, x, residual=r)
        conv_bn._ON = True
        return out
    tt = event_time(two, 20, 3) * 1e6
    fl = 2.0 * B * H * H * C * O
    tot_f += tf
    tot_t += tt
    print("B%d C%4d O%4d H%3d res=%d applies=%s | fused %7.1f us (%5.0f TF/s) err %.2e | gemm + bn_act %7.1f us err %.2e" % (
        B, C, O, H, res, ok, tf, fl / tf / 1e6, err, tt, err2))
print("sum fused %.1f us, two-launch %.1f us" % (tot_f, tot_t))
