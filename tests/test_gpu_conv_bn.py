"""GPU: the fused 1x1 convolution + eval BatchNorm + identity + ReLU of the bf16 trunk (csrc/gemm1x1_mfma.hip,
ops/conv_bn.py) against a plain fp32 torch reference of the same op (conv2d -> batch_norm(eval) -> add -> relu) and its
autograd gradients -- the parity bar for a floating-point kernel: bf16 operands, fp32 accumulation, one rounding, so
the forward agrees to bf16 resolution (2^-8 of the value range) and the gradients to the bf16 resolution of THEIR
operands.  The scale gradient comes from the weight-gradient fold's row dots (ops/conv_bn.py) -- compared with the fp32
autograd value too, at ordinary AND at tiny / zero / negative gammas (where round 5's xhat-from-the-output form broke)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12))


def _make(cuda, B, C, O, H, W, res, seed, small_gamma=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(C, O, 1, bias=False).to(cuda)
    conv.weight.data = (torch.randn(O, C, 1, 1, generator=g) / C ** 0.5).to(cuda).bfloat16()
    bn = torch.nn.BatchNorm2d(O).to(cuda).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.empty(O).uniform_(0.5, 1.5, generator=g)), bn.bias.copy_(torch.randn(O, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(O, generator=g) * 0.3), bn.running_var.copy_(torch.empty(O).uniform_(0.5, 2, generator=g))
        if small_gamma:                          # log-uniform in [1e-8, 1e-1], random signs, every fifth exactly 0; beta ~ 1
            mag = 10.0 ** torch.empty(O).uniform_(-8, -1, generator=g)
            sign = torch.where(torch.rand(O, generator=g) < 0.3, -1.0, 1.0)
            bn.weight.copy_(mag * sign)
            bn.weight[::5] = 0.0
            bn.bias.copy_(torch.randn(O, generator=g) * 0.5 + 0.7)
    r = torch.randn(B, O, H, W, generator=g).to(cuda).bfloat16().contiguous(memory_format=torch.channels_last) if res else None
    return x, conv, bn, r


@pytest.mark.parametrize("B,C,O,H,W,res,relu", [(4, 256, 128, 64, 64, False, True), (2, 128, 512, 40, 56, True, True),
                                                (2, 512, 1024, 16, 16, False, False), (1, 64, 64, 37, 29, True, True),
                                                (3, 2048, 512, 8, 8, False, True), (1, 64, 32, 5, 3, True, False)])
@pytest.mark.parametrize("small_gamma", [False, True])
def test_fused_conv_bn_act_forward_and_backward(cuda, B, C, O, H, W, res, relu, small_gamma):
    from rs_detection_amd.ops import conv_bn
    x, conv, bn, r = _make(cuda, B, C, O, H, W, res, B * C + O, small_gamma)
    assert conv_bn.conv_bn_act_applies(conv, bn, x, r)
    xg = x.clone().requires_grad_(True)
    rg = r.clone().requires_grad_(True) if res else None
    y = conv_bn.conv_bn_act(conv, bn, xg, residual=rg, relu=relu)
    assert "_Conv1x1BNAct" in type(y.grad_fn).__name__ and y.dtype == torch.bfloat16
    assert y.is_contiguous(memory_format=torch.channels_last)
    # fp32 reference on the same (bf16-valued) operands
    xf = x.float().requires_grad_(True)
    wf = conv.weight.detach().float().requires_grad_(True)
    gf, bf = bn.weight.detach().clone().requires_grad_(True), bn.bias.detach().clone().requires_grad_(True)
    rf = r.float().requires_grad_(True) if res else None
    ref = F.batch_norm(F.conv2d(xf, wf), bn.running_mean, bn.running_var, gf, bf, False, 0.0, bn.eps)
    if res:
        ref = ref + rf
    if relu:
        ref = torch.relu(ref)
    scale = float(ref.abs().max())
    assert float((y.float() - ref).abs().max()) <= 2 ** -8 * scale + 1e-6          # one bf16 rounding of the result
    go = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1)).to(cuda)
    go = go.bfloat16().contiguous(memory_format=torch.channels_last)
    # the gate of the reference = the gate of the fused op (a value within a rounding of 0 may differ): use y's
    ref.backward(go.float() * ((y > 0) == (ref > 0)).float() if relu else go.float())
    y.backward(go)
    assert _rel(xg.grad, xf.grad) <= 1.5e-2, _rel(xg.grad, xf.grad)
    assert _rel(conv.weight.grad, wf.grad) <= 1.5e-2
    assert _rel(bn.bias.grad, bf.grad) <= 1e-2
    # the scale gradient: fp32 row dots of bf16-valued operands -- as tight as the bias gradient, whatever gamma is (round 5:
    # 3e-2 at gamma in [0.5, 1.5] and garbage below 1e-3)
    assert _rel(bn.weight.grad, gf.grad) <= 1e-2, _rel(bn.weight.grad, gf.grad)
    assert float((bn.weight.grad - gf.grad).abs().max()) <= 1e-2 * float(gf.grad.abs().max())        # per channel, gamma == 0 included
    if res:
        assert _rel(rg.grad, rf.grad) <= 1e-2


def test_fused_form_tracks_the_two_launch_form_through_a_bottleneck(cuda):
    """A whole Bottleneck (conv1 / conv3 fused, conv2 through MIOpen + bn_act) against the same block with the fused
    form off: outputs and every gradient agree to bf16 resolution; the fused form really ran."""
    from rs_detection_amd.models.backbones.resnet import Bottleneck
    from rs_detection_amd.ops import conv_bn
    torch.manual_seed(0)
    down = torch.nn.Sequential(torch.nn.Conv2d(256, 512, 1, bias=False), torch.nn.BatchNorm2d(512))
    blk = Bottleneck(256, 128, 1, down).to(cuda)
    for m in blk.modules():
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data = m.weight.data.bfloat16().contiguous(memory_format=torch.channels_last)
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.uniform_(0.5, 1.5), m.bias.normal_(0, 0.2), m.running_mean.normal_(0, 0.2), m.running_var.uniform_(0.5, 2)
    blk.eval()                                   # norm_eval: BatchNorm in eval mode, gradients still flow
    x = torch.randn(2, 256, 48, 40, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    g = torch.randn(2, 512, 48, 40, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    outs = []
    for on in (True, False):
        conv_bn._ON = on
        try:
            xi = x.clone().requires_grad_(True)
            blk.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = blk(xi)
            if on:
                assert "_Conv1x1BNAct" in type(y.grad_fn).__name__
            y.backward(g)
            outs.append([y.detach().float(), xi.grad.float()] + [p.grad.float() for p in blk.parameters()])
        finally:
            conv_bn._ON = True
    # outputs: bf16 resolution.  Gradients: the two forms round the pre-activation differently, so the ReLU gates of the
    # few elements within a rounding of zero differ -- each such element moves its gradient by 100 % (measured: 6 % of the
    # gradient norm for ~0.4 % flipped gates); the per-operation test above compares gradients under ONE gate
    assert _rel(outs[0][0], outs[1][0]) <= 3e-2
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert _rel(a, b) <= 0.12, _rel(a, b)


def test_forked_block_outputs_sum_their_gradients_inside_the_batchnorm_backward(cuda):
    """fp32 channels_last ResNet-50: a block's output handed on as a Forked pair (ops/bn_act.py: one autograd output for the
    next block's conv1, one for its identity branch; rsdet_bn_act_backward_nhwc_mask2_f32 sums the two gradients while
    reading) gives the outputs and every gradient of the one-output form, where autograd adds them in a pass of its own."""
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.ops import bn_act as B
    from rs_detection_amd.utils.registry import BACKBONES, build_from_cfg
    torch.manual_seed(0)
    net = build_from_cfg(dict(type="Resnet50", frozen_stages=1, return_stages=["layer1", "layer2", "layer3", "layer4"]),
                         BACKBONES).to(cuda).train()
    net.set_channels_last(True)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_var.uniform_(0.5, 1.5), m.running_mean.normal_(0, 0.2), m.weight.normal_(1, 0.2)
    x = torch.randn(2, 3, 128, 128, device=cuda)
    res = []
    for fork in (True, False):
        B._FORK = fork
        try:
            net.zero_grad(set_to_none=True)
            outs = net(x)
            sum((o * o).mean() * (i + 1) for i, o in enumerate(outs)).backward()
        finally:
            B._FORK = True
        res.append(([o.detach().clone() for o in outs], {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
    (oa, ga), (ob, gb) = res
    for a, b in zip(oa, ob):          # (two forward passes of the same weights: MIOpen may pick another solver the second time)
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())
    assert set(ga) == set(gb) and len(ga) > 100
    # (5e-3 of the L2 norm: the two forward passes already differ by solver round-off, a few ReLU gates flip on it, and each
    #  moves the gradients behind it -- 1.0e-3 measured on the worst parameter in a full-suite run, < 1e-3 in isolation)
    for n in ga:
        d = float((ga[n] - gb[n]).norm())
        assert d <= 5e-3 * float(gb[n].norm()) + 1e-12, (n, d)
