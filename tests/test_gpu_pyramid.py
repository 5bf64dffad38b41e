"""GPU: the pyramid canvas (csrc/canvas.hip, ops/pyramid.py) and the S2ANet head's canvas path against the reference's
per-level loop (/root/reference/python/jdet/models/roi_heads/s2anet_head.py:207-255 -- forward_single under
multi_apply, kept in the product as S2ANetHead.forward_single).

pack / unpack are copies: bit-exact against torch slicing, both directions, both memory formats, 2- and 4-byte elements,
ragged (non-square, odd) level sizes.  The head: same weights through forward_packed and through the level loop ->
the same 25 maps and the same gradients (every parameter + every FPN input), to convolution round-off in fp32."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SIZES = [[(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)],
         [(25, 38), (13, 19), (7, 10), (4, 5), (2, 3)],           # a 200 x 300 image: odd, non-square, ragged
         [(16, 16), (8, 8)],
         [(5, 40), (3, 20), (2, 10), (1, 5), (1, 3), (1, 2), (1, 1), (1, 1)]]


def _slice_pack(levels, lay):
    B, C = levels[0].shape[:2]
    c = torch.zeros((B, C, lay.Hc, lay.Wc), dtype=levels[0].dtype, device=levels[0].device)
    for t, (y0, x0, h, w) in zip(levels, lay.rects):
        c[:, :, y0:y0 + h, x0:x0 + w] = t
    return c


@pytest.mark.parametrize("sizes", SIZES)
@pytest.mark.parametrize("dtype,C", [(torch.float32, 16), (torch.bfloat16, 8), (torch.float32, 5), (torch.bfloat16, 15)])
def test_pack_unpack_are_exact_copies(cuda, sizes, dtype, C):
    from rs_detection_amd.ops.pyramid import canvas_layout, pyramid_pack, pyramid_unpack
    lay = canvas_layout(sizes, cuda)
    assert (lay.Hc * lay.Wc) % 4 == 0 and 0.25 < lay.fill <= 1.0
    # levels never touch or overlap: every pair of rects is separated by at least one gap pixel
    pm = lay.pixmap.view(lay.Hc, lay.Wc).cpu().numpy()
    lvl = np.where(pm >= 0, pm >> 27, -1)
    for dy, dx in ((0, 1), (1, 0), (1, 1), (1, -1)):
        a = lvl[max(dy, 0):, max(dx, 0):lvl.shape[1] + min(dx, 0)]
        b = lvl[:lvl.shape[0] - dy, max(-dx, 0):lvl.shape[1] - max(dx, 0)]
        assert not ((a >= 0) & (b >= 0) & (a != b)).any()
    g = torch.Generator(device="cpu").manual_seed(len(sizes) * 100 + C)
    B = 3
    levels = [torch.randn((B, C, h, w), generator=g).to(cuda).to(dtype) for h, w in sizes]
    want = _slice_pack(levels, lay)
    for cl_in in (False, True):
        lv = [t.contiguous(memory_format=torch.channels_last) for t in levels] if cl_in else levels
        for cl_canvas in (False, True):
            canvas = pyramid_pack(lv, lay, channels_last=cl_canvas)
            assert canvas.is_contiguous(memory_format=torch.channels_last if cl_canvas else torch.contiguous_format)
            assert torch.equal(canvas, want), (cl_in, cl_canvas)
            for cl_out in (False, True):
                back = pyramid_unpack(canvas, lay, channels_last=cl_out)
                assert all(torch.equal(a, b) for a, b in zip(back, levels)), (cl_in, cl_canvas, cl_out)


def test_pack_unpack_gradients(cuda):
    from rs_detection_amd.ops.pyramid import canvas_layout, pyramid_pack, pyramid_unpack
    sizes = SIZES[1]
    lay = canvas_layout(sizes, cuda)
    torch.manual_seed(3)
    levels = [torch.randn((2, 8, h, w), device=cuda, requires_grad=True) for h, w in sizes]
    wc = torch.randn((2, 8, lay.Hc, lay.Wc), device=cuda)
    (pyramid_pack(levels, lay) * wc).sum().backward()
    for t, (y0, x0, h, w) in zip(levels, lay.rects):
        assert torch.equal(t.grad, wc[:, :, y0:y0 + h, x0:x0 + w])
    canvas = torch.randn((2, 8, lay.Hc, lay.Wc), device=cuda, requires_grad=True)
    out = pyramid_unpack(canvas, lay)
    ws = [torch.randn_like(o) for o in out]
    sum((o * w).sum() for o, w in list(zip(out, ws))[:-1]).backward()      # the last level unused: its gradient is zero
    want = _slice_pack(ws[:-1] + [torch.zeros_like(ws[-1])], lay)
    assert torch.equal(canvas.grad, want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cl", [False, True])
@pytest.mark.parametrize("relu", [True, False])
def test_canvas_bias_act_matches_torch(cuda, dtype, cl, relu):
    from rs_detection_amd.ops.pyramid import canvas_layout, canvas_bias_act
    lay = canvas_layout(SIZES[1], cuda)
    torch.manual_seed(11)
    C = 32
    x = torch.randn((2, C, lay.Hc, lay.Wc), device=cuda).to(dtype)
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    bias = torch.randn(C, device=cuda, requires_grad=True)
    y = canvas_bias_act(x, bias, lay, relu)
    ref = x.detach().float() + bias.detach()[None, :, None, None]
    ref = (torch.relu(ref) if relu else ref) * lay.live_f
    assert torch.equal(y.float(), ref.to(dtype).float())
    assert (y[:, :, lay.live_f[0, 0] == 0] == 0).all()
    gy = torch.randn_like(y)
    y.backward(gy)
    gate = (ref > 0).float() if relu else lay.live_f.expand_as(ref)
    gx = gy.float() * gate
    assert torch.equal(x.grad.float(), gx.to(dtype).float())
    np.testing.assert_allclose(bias.grad.cpu().numpy(), gx.sum((0, 2, 3)).cpu().numpy(), rtol=2e-3 if dtype == torch.bfloat16 else 1e-5,
                               atol=1e-4)


def _head(cuda, all_positive=False):
    """``all_positive``: tower biases of +3 under the shipped small weights keep every pre-activation positive, so no
    ReLU gate can flip between the two paths on a 1e-6 difference -- gradients then agree to round-off everywhere, and
    the gap pixels (also positive before masking) exercise the canvas epilogue's zeroing.  Otherwise: weights x 4 so the
    ReLUs bite; a handful of gates flip, so gradients are compared in the L2 norm."""
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.config import Config
    from rs_detection_amd.utils.registry import HEADS, build_from_cfg
    cfg = Config(os.path.join(ROOT, "configs", "s2anet", "s2anet_r50_fpn_1x_dota.py"))
    torch.manual_seed(0)
    head = build_from_cfg(cfg.model["bbox_head"], HEADS).to(cuda)
    with torch.no_grad():
        if all_positive:
            for m in list(head.fam_reg_convs) + list(head.fam_cls_convs) + list(head.odm_reg_convs) + list(head.odm_cls_convs):
                m.conv.bias.fill_(3.0)
                m.conv.weight.mul_(0.1)
            head.align_conv.deform_conv.weight.abs_().mul_(0.1)      # positive inputs (the test's) -> positive outputs
            head.or_conv.bias.fill_(1.0)
        else:
            for p in head.parameters():
                if p.requires_grad and p.dim() > 1:
                    p.mul_(4.0)
    return head


def _run(head, feats, packed, train):
    head.train(train)
    for p in head.parameters():
        p.grad = None
    xs = [f.clone().requires_grad_(True) for f in feats]
    if packed:
        head.canvas_groups = packed
        try:
            outs = head.forward_levels(xs)
        finally:
            head.canvas_groups = None
    else:
        outs = tuple(map(list, zip(*[head.forward_single(x, s) for x, s in zip(xs, head.anchor_strides)])))
    maps = [m for group in outs for m in group if m is not None]
    if train:
        torch.manual_seed(5)
        loss = sum((m.float() * torch.randn_like(m.float())).sum() for m in maps if m.requires_grad)
        loss.backward()
    grads = {n: p.grad.clone() for n, p in head.named_parameters() if p.grad is not None}
    return maps, [x.grad for x in xs], grads


def _rel_l2(a, b):
    return float((a.detach() - b.detach()).norm()) / (float(a.detach().norm()) + 1e-12)


@pytest.mark.parametrize("sizes", [SIZES[0], SIZES[1]])
@pytest.mark.parametrize("train,all_positive", [(True, True), (True, False), (False, False)])
@pytest.mark.parametrize("groups", ["all", "split"])
def test_head_canvas_path_equals_the_level_loop_fp32(cuda, sizes, train, all_positive, groups):
    head = _head(cuda, all_positive)
    torch.manual_seed(1)
    B = 2
    feats = [torch.randn((B, 256, h, w), device=cuda) for h, w in sizes]
    if all_positive:
        feats = [f.abs() * 0.5 for f in feats]
    assert head._packed_ok(feats)
    with torch.set_grad_enabled(train):
        m_loop, gx_loop, gp_loop = _run(head, feats, False, train)
        m_can, gx_can, gp_can = _run(head, feats, groups, train)
    assert len(m_loop) == len(m_can) == (25 if train else 20)
    for a, b in zip(m_loop, m_can):
        assert a.shape == b.shape
        scale = float(a.detach().abs().max()) + 1e-6
        assert float((a.detach() - b.detach()).abs().max()) <= 2e-4 * scale, (a.shape, scale)
    if train:
        assert set(gp_loop) == set(gp_can) and len(gp_loop) >= 26
        # all_positive: no ReLU gate can flip -> agreement to round-off at every element.  Otherwise some gates flip on
        # 1e-7 differences of their pre-activations and each moves the gradients of its (six layers deep) receptive
        # field: the L2 distance stays small, single elements do not
        for a, b in zip(gx_loop, gx_can):
            scale = float(a.abs().max())
            assert scale > 0
            # (6e-3, was 3e-3: in isolation this case measures 9.7e-4 on the input gradients and 2.6e-3 on the parameter
            #  gradients -- 88 % of the old bound -- and MIOpen's fp32 solver choice for the canvas and level shapes is not
            #  the same in every process (find mode FAST falls back by workspace availability): one full-suite run in
            #  three of round 6 failed here, the same code passed before and after)
            if all_positive:
                assert float((a - b).abs().max()) <= 6e-3 * scale      # sums of positive terms: cancellation
            else:
                assert _rel_l2(a, b) <= 5e-2
        for n in gp_loop:
            assert _rel_l2(gp_loop[n], gp_can[n]) <= (6e-3 if all_positive else 5e-2), n


def test_head_canvas_path_bf16_channels_last(cuda):
    """The bf16 step's form: channels_last bf16 maps under autocast.  One canvas convolution and five level
    convolutions go to different MIOpen solvers (some round their bf16 outputs to nearest, the assembly implicit-GEMM
    ones truncate), so the two bf16 results differ by bf16 round-off -- the yardstick is the fp32 result of the same
    weights: the canvas path's distance from it stays within 2.5 x the level loop's, map by map."""
    head = _head(cuda, False)
    with torch.no_grad():
        for p in head.parameters():
            if p.requires_grad and p.dim() > 1:
                p.mul_(0.5)                       # x 2 over the shipped init
    torch.manual_seed(2)
    feats = [torch.randn((2, 256, h, w), device=cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
             for h, w in SIZES[0]]
    truth, _, gt = _run(head, [f.float().contiguous() for f in feats], False, True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        m_loop, _, gp_loop = _run(head, feats, False, True)
        m_all, _, gp_all = _run(head, feats, "all", True)
        m_split, _, gp_split = _run(head, feats, "split", True)
    for m_can, gp_can in ((m_all, gp_all), (m_split, gp_split)):
        for t, a, b in zip(truth, m_loop, m_can):
            t, a, b = t.detach().float(), a.detach().float(), b.detach().float()
            e_loop, e_can = float((a - t).abs().mean()), float((b - t).abs().mean())
            assert e_can <= 2.5 * e_loop + 1e-6 * float(t.abs().mean()), (tuple(t.shape), e_loop, e_can)
        for n in gp_loop:
            e_loop, e_can = _rel_l2(gt[n], gp_loop[n].float()), _rel_l2(gt[n], gp_can[n].float())
            assert e_can <= 2.5 * e_loop + 1e-3, (n, e_loop, e_can)


def test_model_train_step_uses_the_canvas(cuda, monkeypatch):
    """The whole S2ANet model: the step goes through forward_packed by default and through the loop with
    ``bbox_head.packed = False``, same losses."""
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.config import Config
    from rs_detection_amd.utils.registry import MODELS, build_from_cfg
    from rs_detection_amd.utils.synthetic import synthetic_targets
    cfg = Config(os.path.join(ROOT, "configs", "s2anet", "s2anet_r50_fpn_1x_dota.py"))
    torch.manual_seed(0)
    model = build_from_cfg(cfg.model, MODELS).to(cuda).train()
    imgs = torch.randn((2, 3, 256, 256), device=cuda)
    targets = synthetic_targets(2, img=256)
    calls = []
    orig = type(model.bbox_head).forward_packed
    monkeypatch.setattr(type(model.bbox_head), "forward_packed", lambda self, f, **kw: calls.append(1) or orig(self, f, **kw))
    out_c = model(imgs, targets)
    assert calls == [1]
    monkeypatch.setattr(model.bbox_head, "packed", False)
    out_l = model(imgs, targets)
    assert calls == [1]
    for k in out_l:
        a = torch.stack(list(out_l[k])) if isinstance(out_l[k], (list, tuple)) else out_l[k]
        b = torch.stack(list(out_c[k])) if isinstance(out_c[k], (list, tuple)) else out_c[k]
        assert torch.allclose(a, b, rtol=2e-3, atol=1e-5), (k, a, b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cl", [False, True])
def test_orientation_maxpool_matches_amax(cuda, dtype, cl):
    """RotationInvariantPooling (orn.py:595-617) through csrc/arf.hip == torch.amax over the orientation axis, values
    and gradients, with ties (bf16 makes them common; zeros at canvas gaps make them certain)."""
    from rs_detection_amd.ops.orn import RotationInvariantPooling
    torch.manual_seed(4)
    x = torch.randn((3, 64, 9, 13), device=cuda).to(dtype)
    x[:, :, 2:4] = 0                                     # all-equal groups
    x[:, 8:16, 5] = x[:, 8:9, 5]                         # one tied group per pixel of row 5
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    pool = RotationInvariantPooling(64, 8).to(cuda)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y = pool(xa)
    ref = xb.view(3, 8, 8, 9, 13).amax(dim=2)
    assert y.shape == ref.shape and torch.equal(y, ref)
    g = torch.randn_like(ref)
    y.backward(g)
    ref.backward(g)
    assert torch.allclose(xa.grad.float(), xb.grad.float(), rtol=1e-2 if dtype == torch.bfloat16 else 1e-6, atol=1e-6)
    assert ((xa.grad != 0) == (xb.grad != 0)).all()


def test_model_eval_detections_canvas_equals_loop(cuda, monkeypatch):
    """Inference through the whole model (backbone -> FPN -> head -> decode -> NMS): the canvas head returns the
    detections of the level loop -- same count, same labels, scores and polygons to fp32 convolution round-off."""
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.config import Config
    from rs_detection_amd.utils.registry import MODELS, build_from_cfg
    from rs_detection_amd.utils.synthetic import synthetic_targets
    cfg = Config(os.path.join(ROOT, "configs", "s2anet", "s2anet_r50_fpn_1x_dota.py"))
    torch.manual_seed(0)
    model = build_from_cfg(cfg.model, MODELS).to(cuda).eval()
    with torch.no_grad():
        model.bbox_head.odm_cls.bias.fill_(-1.0)          # random weights: let a few hundred anchors pass score_thr
    torch.manual_seed(3)
    imgs = torch.randn((2, 3, 320, 256), device=cuda)
    targets = synthetic_targets(2, img=256)
    for t in targets:
        t["img_size"], t["pad_shape"] = (256, 320), (256, 320)
    with torch.no_grad():
        det_c = model(imgs, targets)
        monkeypatch.setattr(model.bbox_head, "packed", False)
        det_l = model(imgs, targets)
    assert len(det_c) == len(det_l) == 2
    n = 0
    for (pc, sc, lc), (pl, sl, ll) in zip(det_c, det_l):
        assert pc.shape[0] > 0 and abs(pc.shape[0] - pl.shape[0]) <= max(2, pl.shape[0] // 100)
        # random weights give hundreds of near-tied scores: the order of the lists (and a few NMS decisions) flip on 1e-6
        # differences, so match the two sets -- same label, score within 1e-4, polygon within 2e-3 px -- both ways
        same = (lc[:, None] == ll[None, :]) & ((sc[:, None] - sl[None, :]).abs() <= 1e-4 + 1e-4 * sl[None, :].abs()) \
            & ((pc[:, None, :] - pl[None, :, :]).abs().amax(-1) <= 2e-3 + 1e-4 * pl[None].abs().amax(-1))
        assert float(same.any(1).float().mean()) >= 0.99 and float(same.any(0).float().mean()) >= 0.99
        n += pc.shape[0]
    assert n > 20


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,O,H,W,bias", [(2, 64, 256, 40, 24, False), (1, 256, 64, 300, 260, True), (2, 512, 256, 16, 16, True)])
def test_conv1x1_gemm_split_matches_conv2d(cuda, dtype, B, C, O, H, W, bias):
    """ops/conv1x1.py: a 1x1 / stride-1 convolution of a channels_last map as GEMMs on views (forward for small maps,
    backward-data always) + MIOpen's weight gradient == nn.Conv2d, values and all three gradients; both sides of the
    forward's pixel-count switch (300 x 260 > 65 536 pixels)."""
    from rs_detection_amd.ops.conv1x1 import conv1x1, conv1x1_applies
    torch.manual_seed(B * C + O)
    conv = torch.nn.Conv2d(C, O, 1, bias=bias).to(cuda).to(memory_format=torch.channels_last)
    x = torch.randn((B, C, H, W), device=cuda).contiguous(memory_format=torch.channels_last)
    g = torch.randn((B, O, H, W), device=cuda).contiguous(memory_format=torch.channels_last)
    assert conv1x1_applies(conv, x) and not conv1x1_applies(conv, x.contiguous())
    assert not conv1x1_applies(torch.nn.Conv2d(C, O, 1, stride=2).to(cuda), x)
    outs = []
    for fn in (lambda t: conv1x1(conv, t), lambda t: conv(t)):
        xa = x.clone().requires_grad_(True)
        conv.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            y = fn(xa)
        assert y.dtype == dtype and y.is_contiguous(memory_format=torch.channels_last)
        y.backward(g.to(dtype))
        outs.append((y.detach().float(), xa.grad.float(), conv.weight.grad.float().clone(),
                     conv.bias.grad.float().clone() if bias else None))
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    for a, b in zip(*outs):
        if a is None:
            continue
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()) + 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,H,W,bias", [(2, 64, 40, 24, False), (1, 32, 37, 51, True), (2, 256, 16, 20, True)])
def test_conv3x3_backward_data_through_the_forward_solver(cuda, monkeypatch, dtype, B, C, H, W, bias):
    """ops/conv3x3.py: a square 3x3 / stride-1 / padding-1 convolution whose input gradient is computed as a FORWARD
    convolution of the output gradient with the flipped, channel-transposed weights == nn.Conv2d: values and all three
    gradients (odd sizes: the borders are where a wrong flip or padding would show)."""
    from rs_detection_amd.ops import conv3x3 as c3
    from rs_detection_amd.ops.conv3x3 import fast_conv, conv3x3_applies
    monkeypatch.setattr(c3, "_F32", True)         # the fp32 route is off by default (neutral on the step): test it too
    torch.manual_seed(C + H)
    conv = torch.nn.Conv2d(C, C, 3, padding=1, bias=bias).to(cuda).to(memory_format=torch.channels_last)
    x = torch.randn((B, C, H, W), device=cuda).to(dtype).contiguous(memory_format=torch.channels_last)
    g = torch.randn((B, C, H, W), device=cuda).contiguous(memory_format=torch.channels_last)
    xr = x.clone().requires_grad_(True)
    assert conv3x3_applies(xr, conv.weight) and not conv3x3_applies(x, conv.weight)          # no gradient wanted: plain
    assert not conv3x3_applies(xr, torch.nn.Conv2d(C, 2 * C, 3, padding=1).to(cuda).weight)   # not square
    assert not conv3x3_applies(xr.detach().contiguous().requires_grad_(True), conv.weight)    # NCHW
    outs = []
    for fn in (lambda t: fast_conv(conv, t), lambda t: conv(t)):
        xa = x.clone().requires_grad_(True)
        conv.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            y = fn(xa)
        assert y.dtype == dtype
        y.backward(g.to(dtype))
        outs.append((y.detach().float(), xa.grad.float(), conv.weight.grad.float().clone(),
                     conv.bias.grad.float().clone() if bias else None))
    tol = 2e-2 if dtype == torch.bfloat16 else 3e-5
    for a, b in zip(*outs):
        if a is None:
            continue
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()) + 1e-6


def test_orconv_backward_data_through_the_forward_solver(cuda, monkeypatch):
    """ORConv2d (8 rotated copies of the filters: 256 -> 256 as a plain 3x3 convolution of the rotated weight) takes the
    same route; gradients reach the ARF weight through rotate_arf as before."""
    from rs_detection_amd.ops.orn import ORConv2d
    from rs_detection_amd.ops import conv3x3 as c3
    monkeypatch.setattr(c3, "_F32", True)
    torch.manual_seed(7)
    m = ORConv2d(64, 8, kernel_size=3, padding=1, arf_config=(1, 8)).to(cuda)
    x = torch.randn((2, 64, 20, 24), device=cuda).contiguous(memory_format=torch.channels_last)
    g = torch.randn((2, 64, 20, 24), device=cuda)
    res = []
    for cl in (True, False):
        xa = (x if cl else x.contiguous()).clone().requires_grad_(True)
        m.zero_grad()
        y = m(xa)
        y.backward(g)
        res.append((y.detach(), xa.grad.clone(), m.weight.grad.clone(), m.bias.grad.clone()))
    for a, b in zip(*res):
        assert float((a - b).abs().max()) <= 3e-5 * float(b.abs().max()) + 1e-6


def _conv_truth(x, w, bias, live, relu=True):
    """fp32 convolution of the bf16-valued operands (exact products, fp32 sums) + bias, ReLU, gap mask."""
    y = torch.nn.functional.conv2d(x.float().contiguous(), w.float().contiguous(), bias, 1, 1)
    if relu:
        y = torch.relu(y)
    if live is not None:
        y = y * live.view(1, 1, *y.shape[2:]).float()
    return y


@pytest.mark.parametrize("B,C,O,H,W,masked", [(2, 256, 256, 9, 196, True), (1, 64, 32, 3, 224, False),
                                              (1, 128, 256, 4, 230, True), (2, 64, 64, 1, 5, False),
                                              (1, 192, 96, 5, 33, True)])
def test_conv3x3_mfma_bias_relu_mask_forward_backward(cuda, B, C, O, H, W, masked):
    """csrc/conv3x3_mfma.hip behind ops/conv3x3._Conv3x3BiasReLU: relu(conv + bias) with the gap pixels zeroed, in one
    launch -- against the fp32 convolution of the same bf16 operands.  One bf16 rounding of an fp32 sum: within one bf16
    step (2^-8 relative) of the rounded truth up to the summation order; the masked positions are exact zeros.  Ragged
    rows (W not a multiple of the 224-position tile, a second tile of 6 positions, H = 1) exercise the halo and tail
    logic; the gradients: the same three a Conv2d + ReLU pair gives, gate taken from the kernel's own output."""
    from rs_detection_amd.ops import conv3x3 as c3
    torch.manual_seed(B * 1000 + W)
    x = torch.randn((B, C, H, W), device=cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((O, C, 3, 3), device=cuda) / (3 * C ** 0.5)).to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    bias = torch.randn((O,), device=cuda) * 0.2
    live = None
    if masked:
        live = (torch.rand((H * W,), device=cuda) > 0.2).to(torch.uint8)
        x = (x * live.view(1, 1, H, W).to(x.dtype)).contiguous(memory_format=torch.channels_last)
    xa, wa, ba = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    y = c3._Conv3x3BiasReLU.apply(xa, wa, ba, live)
    assert y.dtype == torch.bfloat16 and y.shape == (B, O, H, W) and y.is_contiguous(memory_format=torch.channels_last)
    truth = _conv_truth(x, w, bias, live)
    err = (y.detach().float() - truth).abs()
    assert float((err - truth.abs() * 2.0 ** -7).max()) <= 1e-3 * float(truth.abs().max()), float(err.max())
    if masked:
        dead = (live == 0).view(1, 1, H, W).expand_as(y)
        assert float(y.detach()[dead].abs().max()) == 0.0
    g = torch.randn((B, O, H, W), device=cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y.backward(g)
    # reference gradients: fp32 autograd through conv2d with the ReLU gate of the kernel's own y (no gate flips at y~0)
    xr, wr, br = x.float().contiguous().requires_grad_(True), w.float().contiguous().requires_grad_(True), \
        bias.clone().requires_grad_(True)
    pre = torch.nn.functional.conv2d(xr, wr, br, 1, 1)
    pre.backward(g.float() * (y.float() > 0))
    for name, a, b, tol in (("gx", xa.grad, xr.grad, 1e-2), ("gw", wa.grad, wr.grad, 1e-2), ("gb", ba.grad, br.grad, 2e-3)):
        assert a.shape == b.shape, name
        rel = float((a.float() - b).norm() / b.norm().clamp_min(1e-12))
        assert rel <= tol, (name, rel)


def test_convmodule_takes_the_mfma_kernel_on_the_head_canvas(cuda, monkeypatch):
    """ConvModule(256, 256, 3) + ReLU on a 196-wide bf16 channels_last canvas (the 1024^2 tile's): the one-launch kernel
    runs, and the tower it builds equals the MIOpen + canvas_bias_act sequence it replaces within bf16 round-off --
    yardstick: both against the fp32 tower."""
    from rs_detection_amd.models.utils.modules import ConvModule
    from rs_detection_amd.ops import conv3x3 as c3
    from rs_detection_amd.ops.pyramid import canvas_layout
    lay = canvas_layout([(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)], cuda)
    assert lay.Wc == 196
    torch.manual_seed(5)
    tower = [ConvModule(256, 256, 3, stride=1, padding=1).to(cuda).to(memory_format=torch.channels_last)
             for _ in range(2)]
    x = (torch.randn((1, 256, lay.Hc, lay.Wc), device=cuda) * lay.live_f).to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    calls = []
    real = c3._mfma_conv
    monkeypatch.setattr(c3, "_mfma_conv", lambda *a: (calls.append(1), real(*a))[1])

    def run(on):
        monkeypatch.setattr(c3, "_MFMA", on)
        xa = x.clone().requires_grad_(True)
        for m in tower:
            m.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            t = xa
            for m in tower:
                t = m(t, canvas=lay)
        t.float().square().sum().backward()
        return [t.detach().float(), xa.grad.float()] + [m.conv.weight.grad.float().clone() for m in tower] + \
            [m.conv.bias.grad.float().clone() for m in tower]

    ours = run(True)
    assert len(calls) == 4                                   # 2 forward + 2 backward-data launches
    theirs = run(False)
    assert len(calls) == 4
    xf = x.float().contiguous().requires_grad_(True)
    t = xf
    for m in tower:
        m.zero_grad()
        t = torch.relu(torch.nn.functional.conv2d(t, m.conv.weight.float().contiguous(), m.conv.bias, 1, 1)) * lay.live_f
    t.square().sum().backward()
    truth = [t.detach(), xf.grad] + [m.conv.weight.grad.float().clone() for m in tower] + \
        [m.conv.bias.grad.float().clone() for m in tower]
    gaps = (lay.live == 0).view(1, 1, lay.Hc, lay.Wc).expand_as(ours[0])
    assert float(ours[0][gaps].abs().max()) == 0.0
    for a, b, c in zip(ours, theirs, truth):
        e_ours = float((a - c).norm() / c.norm())
        e_theirs = float((b - c).norm() / c.norm())
        assert e_ours <= 1.5 * e_theirs + 2e-3, (tuple(c.shape), e_ours, e_theirs)


@pytest.mark.parametrize("B,C,O,H,W", [(2, 256, 256, 9, 196), (1, 64, 32, 5, 37), (2, 128, 96, 9, 300), (1, 256, 256, 3, 224),
                                       (2, 64, 256, 17, 1), (3, 64, 8, 2, 33), (1, 64, 264, 4, 40)])
@pytest.mark.parametrize("out_bf16", [False, True])
def test_conv3x3_wrw_mfma_equals_the_fp32_weight_gradient(cuda, B, C, O, H, W, out_bf16):
    """csrc/conv3x3_wrw_mfma.hip through its C ABI: dW of a 3x3 / stride 1 / padding 1 convolution from bf16 channels-last
    g and x -- against the fp32 weight gradient of the same operands.  bf16 products are exact in fp32 and the sums are
    fp32: the fp32 output agrees to summation order (1e-5 of the largest entry), the bf16 output to one rounding.  Rows
    shorter / longer than the 32-position chunk, a single column, O below / above one 256-channel block, row groups
    of unequal size."""
    from rs_detection_amd import _lib
    lib = _lib.load()
    torch.manual_seed(B * 100 + W)
    x = torch.randn(B, C, H, W, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, O, H, W, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    assert lib.rsdet_conv3x3_wrw_mfma_supported(B, H, W, C, O)
    gw = torch.empty((O, C, 3, 3), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=cuda,
                     memory_format=torch.channels_last)
    nb = lib.rsdet_conv3x3_wrw_mfma_ws_size(B, H, W, C, O)
    ws = torch.empty((nb,), dtype=torch.uint8, device=cuda)
    _lib.check(lib.rsdet_conv3x3_wrw_mfma_bf16(_lib.ptr(g), _lib.ptr(x), B, H, W, C, O, _lib.ptr(gw), int(out_bf16),
                                               _lib.ptr(ws), nb, _lib.stream_ptr()), "rsdet_conv3x3_wrw_mfma_bf16")
    w0 = torch.zeros(O, C, 3, 3, device=cuda)
    ref = torch.ops.aten.convolution_backward(g.float().contiguous(), x.float().contiguous(), w0, None, (1, 1), (1, 1), (1, 1),
                                              False, (0, 0), 1, (False, True, False))[1]
    err = float((gw.float() - ref).abs().max() / ref.abs().max())
    assert err <= (4e-3 if out_bf16 else 1e-5), err


def test_two_layer_tower_node_equals_the_per_layer_nodes(cuda):
    """S2ANetHead._tower: two stacked conv + bias + ReLU ConvModules on the bf16 canvas as ONE autograd node
    (ops/conv3x3._Conv3x3Tower2: the first layer's ReLU gate and bias-gradient sums ride in the epilogue of the second
    layer's backward-data) against the same two layers as two nodes.  The gate is read from the same stored bf16 tensor and
    the backward-data arithmetic is the same kernel's, so outputs, input gradient and both weight gradients are
    BIT-identical; the first layer's bias gradient differs by the bf16 rounding of its addends (4e-3 relative)."""
    from rs_detection_amd.models.roi_heads.s2anet_head import S2ANetHead
    from rs_detection_amd.models.utils.modules import ConvModule
    from rs_detection_amd.ops import conv3x3 as c3
    from rs_detection_amd.ops.pyramid import canvas_layout
    lay = canvas_layout([(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)], cuda)
    torch.manual_seed(11)
    tower = torch.nn.ModuleList([ConvModule(256, 256, 3, stride=1, padding=1) for _ in range(2)]).to(cuda)
    for m in tower:
        m.conv.weight.data = m.conv.weight.data.bfloat16().contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            m.conv.bias.normal_(0, 0.1)
    x = (torch.randn((2, 256, lay.Hc, lay.Wc), device=cuda) * lay.live_f).to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    go = torch.randn((2, 256, lay.Hc, lay.Wc), device=cuda).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    outs = []
    for on in (True, False):
        c3._TOWER = on
        try:
            xa = x.clone().requires_grad_(True)
            tower.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                t = S2ANetHead._tower(tower, xa, lay)
            assert ("_Conv3x3Tower2" in type(t.grad_fn).__name__) == on
            t.backward(go)
            outs.append([t.detach(), xa.grad] + [m.conv.weight.grad.clone() for m in tower]
                        + [m.conv.bias.grad.clone() for m in tower])
        finally:
            c3._TOWER = True
    one, two = outs
    for i in (0, 1, 2, 3, 5):        # output, grad_x, grad_w1, grad_w2, grad_b2
        assert torch.equal(one[i], two[i]), i
    # (the fused epilogue sums the fp32 values BEFORE their bf16 rounding, the per-layer pass sums the stored bf16 gradient:
    #  ~1e5 roundings of 2^-9 each per channel -- the fused sum is the more accurate of the two)
    assert float((one[4] - two[4]).abs().max()) <= 4e-3 * float(two[4].abs().max())
    gaps = (lay.live == 0).view(1, 1, lay.Hc, lay.Wc).expand_as(one[0])
    assert float(one[0][gaps].float().abs().max()) == 0.0
    # not taken: fp32 canvas, three layers, no gradient mode
    assert not c3.conv3x3_tower2_applies(x.float(), tower[0].conv, tower[1].conv)
