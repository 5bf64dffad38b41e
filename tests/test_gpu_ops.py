"""GPU parity: ARF, deformable conv pieces, ROIAlignRotated_v1, box coder, fused refine+offset."""
import numpy as np
import pytest
import torch

import oracle
from conftest import dota_boxes

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


# ---------------------------------------------------------------- ARF (a12)
@pytest.mark.parametrize("O,I,nOri,nRot,k", [(4, 3, 1, 8, 3), (32, 256, 1, 8, 3), (5, 2, 4, 8, 3), (6, 7, 8, 4, 1)])
def test_arf_forward_backward_exact(cuda, oracle_c, O, I, nOri, nRot, k):
    from rs_detection_amd.ops import arf_forward, arf_backward
    from rs_detection_amd.ops.orn import arf_indices
    rng = np.random.default_rng(O + I)
    idx = arf_indices(nOri, nRot, (k, k)).numpy()
    w = rng.standard_normal((O, I, nOri, k, k)).astype(np.float32)
    out = arf_forward(_t(w, cuda), _t(idx, cuda)).cpu().numpy()
    assert (out == oracle_c.arf_forward(w, idx)).all()  # pure copy: exact
    go = rng.standard_normal(out.shape).astype(np.float32)
    gw = arf_backward(_t(idx, cuda), _t(go, cuda)).cpu().numpy()
    assert (gw == oracle_c.arf_backward(idx, go)).all()  # same summation order: exact


def test_orconv2d_autograd(cuda):
    from rs_detection_amd.ops import ORConv2d
    torch.manual_seed(0)
    m = ORConv2d(16, 4, kernel_size=3, padding=1, arf_config=(1, 8)).to(cuda)
    x = torch.randn(2, 16, 9, 9, device=cuda, requires_grad=True)
    y = m(x)
    assert y.shape == (2, 32, 9, 9)
    y.square().sum().backward()
    # gradient of the ARF expansion == transpose of a gather: check against autograd on an index_select twin
    idx = m.indices.long() - 1  # (1,3,3,8)
    w = m.weight.detach().clone().requires_grad_(True)
    O, I = w.shape[:2]
    wf = w.view(O, I, 9)
    inv = torch.empty(8, 9, dtype=torch.long, device=cuda)
    for k in range(8):
        inv[k, idx.view(9, 8)[:, k]] = torch.arange(9, device=cuda)
    rot = torch.stack([wf[:, :, inv[k]] for k in range(8)], 1).reshape(O * 8, I, 3, 3)
    y2 = torch.nn.functional.conv2d(x.detach(), rot, m.bias, 1, 1)
    torch.testing.assert_close(y2, y.detach(), atol=1e-5, rtol=1e-5)
    y2.square().sum().backward()
    torch.testing.assert_close(w.grad, m.weight.grad, atol=1e-3, rtol=1e-4)


# ---------------------------------------------------------------- deformable conv (a11)
GEOMS = [
    dict(B=2, C=4, H=6, W=5, kh=3, kw=3, ph=1, pw=1, sh=1, sw=1, dh=1, dw=1, dg=1),
    dict(B=1, C=6, H=9, W=11, kh=3, kw=3, ph=1, pw=1, sh=2, sw=2, dh=1, dw=1, dg=2),
    dict(B=3, C=8, H=16, W=16, kh=3, kw=3, ph=2, pw=2, sh=1, sw=1, dh=2, dw=2, dg=1),
    dict(B=2, C=32, H=32, W=32, kh=3, kw=3, ph=1, pw=1, sh=1, sw=1, dh=1, dw=1, dg=1),
]


def _dcn_inputs(g, seed):
    rng = np.random.default_rng(seed)
    Ho, Wo = oracle._COracle.out_hw(g["H"], g["W"], *[g[k] for k in ("kh", "kw", "ph", "pw", "sh", "sw", "dh", "dw")])
    im = rng.standard_normal((g["B"], g["C"], g["H"], g["W"])).astype(np.float32)
    off = (rng.standard_normal((g["B"], g["dg"] * 2 * g["kh"] * g["kw"], Ho, Wo)) * 2.5).astype(np.float32)
    return im, off, Ho, Wo


@pytest.mark.parametrize("gi", range(len(GEOMS)))
def test_deform_im2col_col2im_coord(cuda, oracle_c, gi):
    from rs_detection_amd.ops import deformable_im2col, deformable_col2im, deformable_col2im_coord
    g = GEOMS[gi]
    im, off, Ho, Wo = _dcn_inputs(g, gi)
    k, p, s, d = (g["kh"], g["kw"]), (g["ph"], g["pw"]), (g["sh"], g["sw"]), (g["dh"], g["dw"])
    col = deformable_im2col(_t(im, cuda), _t(off, cuda), k, p, s, d, g["dg"]).cpu().numpy()
    want = oracle_c.deform_im2col(im, off, g, g["dg"]).reshape(col.shape)
    assert np.abs(col - want).max() <= TOL
    rng = np.random.default_rng(99 + gi)
    gcol = rng.standard_normal(col.shape).astype(np.float32)
    gim = deformable_col2im(_t(gcol, cuda), _t(off, cuda), im.shape, k, p, s, d, g["dg"]).cpu().numpy()
    want = oracle_c.deform_col2im(gcol, off, im.shape, g, g["dg"])
    assert np.abs(gim - want).max() <= 1e-4 * max(1.0, np.abs(want).max())  # atomics: order differs
    goff = deformable_col2im_coord(_t(gcol, cuda), _t(im, cuda), _t(off, cuda), k, p, s, d, g["dg"]).cpu().numpy()
    want = oracle_c.deform_col2im_coord(gcol, im, off, g, g["dg"])
    assert np.abs(goff - want).max() <= 1e-4 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("gi", range(len(GEOMS)))
def test_deform_nhwc_kernels_vs_oracle(cuda, oracle_c, gi):
    """Channels-last kernels: same numbers as the reference-layout oracle, transposed."""
    from rs_detection_amd.ops import deformable_im2col_nhwc, deformable_col2im_nhwc
    g = GEOMS[gi]
    im, off, Ho, Wo = _dcn_inputs(g, gi)
    k, p, s, d = (g["kh"], g["kw"]), (g["ph"], g["pw"]), (g["sh"], g["sw"]), (g["dh"], g["dw"])
    B, C, taps = g["B"], g["C"], g["kh"] * g["kw"]
    x = _t(im, cuda).permute(0, 2, 3, 1).contiguous()
    colT = deformable_im2col_nhwc(x, _t(off, cuda), k, p, s, d, g["dg"]).cpu().numpy()
    want = oracle_c.deform_im2col(im, off, g, g["dg"])  # (C*taps, B, Ho, Wo)
    wantT = want.reshape(C, taps, B * Ho * Wo).transpose(2, 1, 0).reshape(B * Ho * Wo, taps * C)
    assert np.abs(colT - wantT).max() <= TOL
    rng = np.random.default_rng(7 + gi)
    gcolT = rng.standard_normal(colT.shape).astype(np.float32)
    gim = deformable_col2im_nhwc(_t(gcolT, cuda), _t(off, cuda), (B, g["H"], g["W"], C), k, p, s, d, g["dg"])
    gcol = gcolT.reshape(B * Ho * Wo, taps, C).transpose(2, 1, 0).reshape(C * taps, B * Ho * Wo)
    want = oracle_c.deform_col2im(gcol, off, im.shape, g, g["dg"])
    got = gim.permute(0, 3, 1, 2).cpu().numpy()
    assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
    if g["dg"] == 1:  # gather form (no floating-point atomics): same numbers, every element written exactly once
        from rs_detection_amd.ops.dcn_v1 import deformable_col2im_gather_nhwc
        gat = deformable_col2im_gather_nhwc(_t(gcolT, cuda), _t(off, cuda), (B, g["H"], g["W"], C), k, p, s, d)
        got = gat.permute(0, 3, 1, 2).cpu().numpy()
        assert np.isfinite(got).all() and np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max())


def test_deform_col2im_gather_level0_shape_matches_scatter(cuda):
    """The step's own shape (B=4, C=256, 128x128: 8x8 tile swizzle, float4 rows, 65 536-pixel scan over 64 chunks) and
    an awkward one (odd sizes: no swizzle, scalar rows, one partial chunk)."""
    from rs_detection_amd.ops.dcn_v1 import deformable_col2im_nhwc, deformable_col2im_gather_nhwc
    torch.manual_seed(5)
    for (B, C, H, W) in ((4, 256, 128, 128), (2, 30, 37, 21)):
        off = torch.randn(B, 18, H, W, device=cuda) * 2.5
        colT = torch.randn(B * H * W, 9 * C, device=cuda)
        a = deformable_col2im_nhwc(colT, off, (B, H, W, C), (3, 3), (1, 1), (1, 1), (1, 1))
        g = deformable_col2im_gather_nhwc(colT, off, (B, H, W, C), (3, 3), (1, 1), (1, 1), (1, 1))
        assert float((a - g).abs().max()) <= 1e-4 * float(a.abs().max())


def test_deform_col2im_shared_index_over_levels_matches_per_call(cuda):
    """GatherIndexPlan: the five pyramid levels of a step (and ten calls: two chunks of the 8-level C struct; odd sizes,
    fp32 and bf16 columns) indexed by ONE build == each call building its own index.  Entries of a pixel are filed in
    the order an atomic counter hands out in both forms, so sums agree to rounding, not bit for bit."""
    from rs_detection_amd.ops.dcn_v1 import GatherIndexPlan, deformable_col2im_gather_nhwc
    torch.manual_seed(11)
    shapes = [(4, 64, 128, 128), (4, 64, 64, 64), (4, 64, 32, 32), (4, 64, 16, 16), (4, 64, 8, 8),
              (2, 30, 37, 21), (1, 8, 5, 3), (2, 16, 1, 9), (3, 4, 2, 2), (1, 12, 40, 40)]
    for dt in (torch.float32, torch.bfloat16):
        plan, calls = GatherIndexPlan(), []
        for (B, C, H, W) in shapes:
            off = (torch.randn(B, 18, H, W, device=cuda) * 2.5).contiguous()
            colT = torch.randn(B * H * W, 9 * C, device=cuda).to(dt)
            calls.append((colT, off, (B, H, W, C), plan.add(off, C, H, W, (3, 3), (1, 1), (1, 1), (1, 1), B)))
        for colT, off, shp, slot in reversed(calls):          # backward order: last level first
            a = deformable_col2im_gather_nhwc(colT, off, shp, (3, 3), (1, 1), (1, 1), (1, 1))
            b = deformable_col2im_gather_nhwc(colT, off, shp, (3, 3), (1, 1), (1, 1), (1, 1), slot=slot)
            assert float((a - b).abs().max()) <= 2e-6 * max(float(a.abs().max()), 1.0), (shp, dt)
            if dt == torch.bfloat16:    # the gradient of a bf16 input stored rounded by the gather itself == cast afterwards
                c = deformable_col2im_gather_nhwc(colT, off, shp, (3, 3), (1, 1), (1, 1), (1, 1), slot=slot,
                                                  out_dtype=torch.bfloat16)
                assert c.dtype == torch.bfloat16 and torch.equal(c, b.to(torch.bfloat16)), shp
        assert sorted(plan.built) == [0, 1]


def test_deform_conv_backward_shared_index_context(cuda):
    """``with shared_gather_index():`` around three DeformConv calls: same gradients as without it, one index build."""
    from rs_detection_amd.ops import dcn_v1
    torch.manual_seed(12)
    w = (torch.randn(16, 32, 3, 3, device=cuda) * 0.1).requires_grad_()
    xs = [torch.randn(2, 32, s, s, device=cuda).contiguous(memory_format=torch.channels_last).requires_grad_()
          for s in (24, 12, 6)]
    offs = [torch.randn(2, 18, s, s, device=cuda) * 2 for s in (24, 12, 6)]
    grads = []
    for shared in (False, True):
        for t in xs + [w]:
            t.grad = None
        if shared:
            with dcn_v1.shared_gather_index() as plan:
                outs = [dcn_v1.deform_conv(x, o, w, 1, 1, 1, 1, 1) for x, o in zip(xs, offs)]
            assert len(plan.calls) == 3 and not plan.built
        else:
            outs = [dcn_v1.deform_conv(x, o, w, 1, 1, 1, 1, 1) for x, o in zip(xs, offs)]
        sum((o * o).sum() for o in outs).backward()
        if shared:
            assert list(plan.built) == [0]
        grads.append([t.grad.clone() for t in xs + [w]])
    for a, b in zip(*grads):
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())


def test_deform_conv_nhwc_path_equals_reference_layout_path(cuda):
    """DeformConv fast path (channels-last) == reference-layout path, forward and both gradients."""
    from rs_detection_amd.ops.dcn_v1 import DeformConvFunction, DeformConvFunctionNHWC
    torch.manual_seed(3)
    x = torch.randn(2, 32, 24, 20, device=cuda)
    off = torch.randn(2, 18, 24, 20, device=cuda) * 2
    w = torch.randn(16, 32, 3, 3, device=cuda) * 0.1
    outs = []
    for fn, extra, fmt in ((DeformConvFunction, (1, 1, 64), torch.contiguous_format),
                           (DeformConvFunctionNHWC, (1,), torch.channels_last)):
        xi = x.clone().requires_grad_(True)
        wi = w.clone().requires_grad_(True)
        y = fn.apply(xi, off, wi, 1, 1, 1, *extra)
        y.backward(torch.ones_like(y) * 0.5 + y.detach() * 0.1)
        outs.append((y.detach(), xi.grad, wi.grad))
    for a, b in zip(*outs):
        torch.testing.assert_close(a.contiguous(), b.contiguous(), atol=2e-4, rtol=1e-4)


def test_deform_conv_zero_offset_is_conv2d(cuda):
    """Independent pin: with zero offsets DeformConv must equal a plain convolution (fwd + grads)."""
    from rs_detection_amd.ops import DeformConv
    torch.manual_seed(1)
    m = DeformConv(16, 24, 3, padding=1).to(cuda)
    x = torch.randn(2, 16, 20, 17, device=cuda, requires_grad=True)
    off = torch.zeros(2, 18, 20, 17, device=cuda)
    y = m(x, off)
    x2 = x.detach().clone().requires_grad_(True)
    w2 = m.weight.detach().clone().requires_grad_(True)
    y2 = torch.nn.functional.conv2d(x2, w2, None, 1, 1)
    torch.testing.assert_close(y, y2, atol=1e-4, rtol=1e-4)
    gy = torch.randn_like(y)
    y.backward(gy)
    y2.backward(gy)
    torch.testing.assert_close(x.grad, x2.grad, atol=1e-3, rtol=1e-4)
    torch.testing.assert_close(m.weight.grad, w2.grad, atol=1e-3, rtol=1e-4)


def test_deform_conv_offset_grad_finite_difference(cuda):
    from rs_detection_amd.ops import deform_conv
    torch.manual_seed(2)
    x = torch.randn(1, 4, 7, 7, device=cuda)
    w = torch.randn(3, 4, 3, 3, device=cuda)
    off = (torch.rand(1, 18, 7, 7, device=cuda) * 0.6 + 0.2).requires_grad_(True)  # stay inside one cell
    y = deform_conv(x, off, w, 1, 1, 1, 1, 1)
    gy = torch.randn_like(y)
    y.backward(gy)
    eps = 1e-2
    for idx in [(0, 0, 3, 3), (0, 7, 2, 5), (0, 17, 6, 6)]:
        o2 = off.detach().clone()
        o2[idx] += eps
        o3 = off.detach().clone()
        o3[idx] -= eps
        fd = ((deform_conv(x, o2, w, 1, 1, 1, 1, 1) - deform_conv(x, o3, w, 1, 1, 1, 1, 1)) * gy).sum() / (2 * eps)
        assert abs(float(fd) - float(off.grad[idx])) <= 2e-2 * max(1.0, abs(float(fd)))


# ---------------------------------------------------------------- RROIAlign (a18)
def _rois(rng, R, N, span):
    b = dota_boxes(rng, R, span, 8, 120, 60)
    return np.concatenate([rng.integers(0, N, (R, 1)).astype(np.float32), b], 1)


@pytest.mark.parametrize("N,C,H,W,R,scale,sr", [(2, 3, 16, 20, 5, 0.25, 2), (1, 8, 32, 32, 17, 0.125, 2),
                                                 (2, 4, 24, 24, 6, 0.25, 0), (2, 16, 64, 64, 40, 1 / 16., 2)])
def test_rroi_align_forward_backward(cuda, oracle_c, N, C, H, W, R, scale, sr):
    from rs_detection_amd.ops import roi_align_rotated_v1
    rng = np.random.default_rng(R)
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    rois = _rois(rng, R, N, W / scale)
    rois[0, 1:3] = [-30, -30]  # partially outside
    rois[1, 3:5] = [0.5, 0.5]  # malformed -> forced to 1x1 (:101-102)
    ft = _t(feat, cuda).requires_grad_(True)
    out = roi_align_rotated_v1(ft, _t(rois, cuda), (7, 7), scale, sr)
    want = oracle_c.rroi_align_v1_forward(feat, rois, (7, 7), scale, sr)
    assert np.abs(out.detach().cpu().numpy() - want).max() <= TOL
    go = rng.standard_normal(want.shape).astype(np.float32)
    out.backward(_t(go, cuda))
    wantg = oracle_c.rroi_align_v1_backward(go, rois, feat.shape, scale, sr)
    assert np.abs(ft.grad.cpu().numpy() - wantg).max() <= 1e-4 * max(1.0, np.abs(wantg).max())


@pytest.mark.parametrize("N,C,H,W,R,scale,sr", [(2, 3, 16, 20, 5, 0.25, 2), (1, 8, 32, 32, 17, 0.125, 0),
                                                 (2, 16, 64, 64, 40, 1 / 16., 2), (1, 256, 64, 64, 64, 0.125, 2)])
def test_rroi_align_v0_forward_backward(cuda, oracle_c, N, C, H, W, R, scale, sr):
    """f4: ROIAlignRotated (ops/roi_align_rotated.py); sr=0 runs the adaptive grid + scatter backward, sr>0 the gather."""
    from rs_detection_amd.ops.roi_align_rotated import ROIAlignRotated
    rng = np.random.default_rng(100 + R)
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    rois = _rois(rng, R, N, W / scale)
    rois[0, 1:3] = [-30, -30]
    rois[1, 3:5] = [0.5, 0.5]
    ft = _t(feat, cuda).requires_grad_(True)
    out = ROIAlignRotated((7, 7), scale, sr)(ft, _t(rois, cuda))
    want = oracle_c.rroi_align_v1_forward(feat, rois, (7, 7), scale, sr, "v0")
    assert np.abs(out.detach().cpu().numpy() - want).max() <= TOL
    go = rng.standard_normal(want.shape).astype(np.float32)
    out.backward(_t(go, cuda))
    wantg = oracle_c.rroi_align_v1_backward(go, rois, feat.shape, scale, sr, "v0")
    assert np.abs(ft.grad.cpu().numpy() - wantg).max() <= 1e-4 * max(1.0, np.abs(wantg).max())


def test_rroi_align_v0_reference_smoke_shape(cuda):
    """The reference's own smoke test (roi_align_rotated.py:332-339): (2,1024,64,64) feature, two RoIs, 1/16 scale,
    default adaptive sampling; output shape and a finite gradient of exp(output)."""
    from rs_detection_amd.ops.roi_align_rotated import ROIAlignRotated
    feat = torch.randn(2, 1024, 64, 64, device=cuda, requires_grad=True)
    roi = torch.tensor([[0, 20, 120, 80, 195.5, 0.3], [1, 23, 56, 200, 300.5, 0.2]], device=cuda)
    out = ROIAlignRotated((7, 7), 1 / 16.)(feat, roi)
    assert tuple(out.shape) == (2, 1024, 7, 7)
    (g,) = torch.autograd.grad(out.exp().sum(), feat)
    assert torch.isfinite(g).all() and float(g.abs().sum()) > 0


def test_rroi_align_grad_sum_property(cuda):
    """Reference's own smoke property: inside RoIs, sum(grad) = R*C*49 for ones."""
    from rs_detection_amd.ops import ROIAlignRotated_v1
    feat = torch.randn(2, 3, 64, 64, device=cuda, requires_grad=True)
    rois = torch.tensor([[0, 300, 400, 200, 100, 0.3], [1, 500, 500, 150, 80, -0.7]], device=cuda)
    out = ROIAlignRotated_v1((7, 7), 1 / 16., 2)(feat, rois)
    out.sum().backward()
    assert abs(float(feat.grad.sum()) - 2 * 3 * 49) < 1e-2


# ---------------------------------------------------------------- FeatureRefine (f4)
def _fr_boxes(rng, N, H, W, stride):
    """Boxes as the reference's own test draws them (fr.py:349-377)."""
    base = 4.0 * stride
    yc, xc = np.meshgrid(stride * np.arange(H), stride * np.arange(W), indexing="ij")
    xc = xc[None] + base * rng.standard_normal((N, H, W))
    yc = yc[None] + base * rng.standard_normal((N, H, W))
    w = base * np.exp(rng.standard_normal((N, H, W)))
    h = base * np.exp(rng.standard_normal((N, H, W)))
    a = -np.pi / 2 * rng.random((N, H, W))
    return np.stack([xc, yc, w, h, a], -1).astype(np.float32)


@pytest.mark.parametrize("N,C,H,W,stride,points", [(2, 16, 32, 32, 8.0, 1), (2, 16, 32, 32, 8.0, 5),
                                                    (1, 7, 13, 29, 16.0, 5), (2, 256, 64, 64, 16.0, 1),
                                                    (1, 256, 128, 128, 8.0, 5)])
def test_feature_refine_forward_backward(cuda, oracle_c, N, C, H, W, stride, points):
    from rs_detection_amd.ops.fr import FR
    rng = np.random.default_rng(7 * H + points)
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    boxes = _fr_boxes(rng, N, H, W, stride)
    ft = _t(feat, cuda).requires_grad_(True)
    out = FR(1.0 / stride, points)(ft, _t(boxes, cuda))
    want = oracle_c.feature_refine_forward(feat, boxes, 1.0 / stride, points)
    assert np.abs(out.detach().cpu().numpy() - want).max() <= TOL
    go = rng.standard_normal(want.shape).astype(np.float32)
    out.backward(_t(go, cuda))
    wantg = oracle_c.feature_refine_backward(go, boxes, 1.0 / stride, points)
    assert np.abs(ft.grad.cpu().numpy() - wantg).max() <= 1e-4 * max(1.0, np.abs(wantg).max())
    # the identity term alone: grad_in - grad_out is the sampled part, zero where no point lands
    lin = (ft.grad - _t(go, cuda)).abs().sum()
    assert torch.isfinite(lin)


@pytest.mark.parametrize("N,C,H,W,stride,points", [(2, 16, 32, 32, 8.0, 5), (1, 256, 128, 128, 8.0, 5),
                                                    (2, 256, 64, 64, 16.0, 1), (1, 512, 19, 23, 16.0, 5),
                                                    (3, 4, 9, 7, 8.0, 5), (1, 12, 16, 16, 8.0, 5)])
def test_feature_refine_channels_last_matches_the_nchw_form(cuda, oracle_c, N, C, H, W, stride, points):
    """A channels_last map runs the NHWC forward (csrc/feature_refine.hip) and the channels-last gather backward with no
    layout turn: output and gradient come back channels_last, the output is BIT-identical to the NCHW kernel's (same
    sums, same order), the gradient within the backward's own tolerance of the oracle.  C = 12 (C / 4 not a power of two)
    is not taken by the NHWC kernel: the op falls to the NCHW form and still answers."""
    from rs_detection_amd import _lib
    from rs_detection_amd.ops.fr import FR
    rng = np.random.default_rng(11 * H + points + C)
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    boxes = _fr_boxes(rng, N, H, W, stride)
    taken = bool(_lib.load().rsdet_feature_refine_forward_nhwc_supported(C))
    assert taken == (C != 12)
    ref = FR(1.0 / stride, points)(_t(feat, cuda), _t(boxes, cuda))
    fc = _t(feat, cuda).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = FR(1.0 / stride, points)(fc, _t(boxes, cuda))
    assert torch.equal(out, ref)
    if taken and C > 1 and H * W > 1:
        assert out.is_contiguous(memory_format=torch.channels_last) and not out.is_contiguous()
    go = rng.standard_normal(feat.shape).astype(np.float32)
    out.backward(_t(go, cuda).contiguous(memory_format=torch.channels_last))
    wantg = oracle_c.feature_refine_backward(go, boxes, 1.0 / stride, points)
    assert np.abs(fc.grad.cpu().numpy() - wantg).max() <= 1e-4 * max(1.0, np.abs(wantg).max())


def test_feature_refine_module_and_linearity(cuda):
    """FeatureRefineModule (fr.py:291-347) steps; FR is linear in the features: FR(a+b) = FR(a) + FR(b)."""
    from rs_detection_amd.ops.fr import FR, FeatureRefineModule
    rng = np.random.default_rng(3)
    strides = [8, 16]
    xs = [torch.randn(2, 32, 1024 // s // 8, 1024 // s // 8, device=cuda, requires_grad=True) for s in strides]
    best = [[_t(_fr_boxes(rng, 1, x.shape[2], x.shape[3], float(s))[0].reshape(-1, 5), cuda)
             for x, s in zip(xs, strides)] for _ in range(2)]
    m = FeatureRefineModule(32, strides).to(cuda)
    outs = m(xs, best)
    assert [tuple(o.shape) for o in outs] == [tuple(x.shape) for x in xs]
    sum(o.square().mean() for o in outs).backward()
    assert all(torch.isfinite(x.grad).all() and float(x.grad.abs().sum()) > 0 for x in xs)
    assert all(p.grad is not None for p in m.parameters())
    a, b = torch.randn(2, 8, 16, 16, device=cuda), torch.randn(2, 8, 16, 16, device=cuda)
    bx = _t(_fr_boxes(rng, 2, 16, 16, 8.0), cuda)
    f = FR(1 / 8., 5)
    assert float((f(a + b, bx) - f(a, bx) - f(b, bx)).abs().max()) <= 1e-4
    with pytest.raises(AssertionError):
        FR(1 / 8., 3)(a, bx)


# ---------------------------------------------------------------- convex_sort (f4)
@pytest.mark.parametrize("nbs,npts,circular", [(1, 8, True), (63, 24, True), (5000, 24, True), (4097, 8, False),
                                               (300, 56, True), (130, 57, True), (3, 200, False)])
def test_convex_sort_vs_oracle(cuda, oracle_c, nbs, npts, circular):
    """Random clouds with random masks, plus lattice points (ties, duplicates, collinear runs): indices bit-exact."""
    from rs_detection_amd.ops.convex_sort import convex_sort
    rng = np.random.default_rng(nbs + npts)
    for kind in ("cloud", "lattice"):
        pts = ((rng.standard_normal((nbs, npts, 2)) * 20) if kind == "cloud"
               else rng.integers(-3, 4, (nbs, npts, 2))).astype(np.float32)
        m = rng.random((nbs, npts)) > 0.3
        got = convex_sort(_t(pts, cuda), torch.from_numpy(m).to(cuda), circular).cpu().numpy()
        want = oracle.np_convex_sort(pts, m, circular)
        assert (got == want).all(), kind


def test_convex_sort_edges_and_poly_iou_shape(cuda, oracle_c):
    from rs_detection_amd.ops.convex_sort import convex_sort
    # empty batch, no points (:179-180 returns the -1 table), everything masked off, one valid point
    assert tuple(convex_sort(torch.zeros(0, 24, 2, device=cuda), torch.zeros(0, 24, device=cuda)).shape) == (0, 25)
    z = convex_sort(torch.zeros(4, 0, 2, device=cuda), torch.zeros(4, 0, device=cuda))
    assert tuple(z.shape) == (4, 1) and (z == -1).all()
    pts = torch.randn(3, 6, 2, device=cuda)
    m = torch.zeros(3, 6, dtype=torch.bool, device=cuda)
    m[1, 4] = True
    got = convex_sort(pts, m).cpu().numpy()
    assert (got == oracle.np_convex_sort(pts.cpu().numpy(), m.cpu().numpy())).all()
    assert got[1, 0] == 4 and got[1, 1] == 4 and (got[1, 2:] == -1).all()
    # the caller's use (poly_iou_loss.py:21-37): area of the intersection of two overlapping squares
    sq1 = np.array([[0, 0], [2, 0], [2, 2], [0, 2]], np.float32)
    sq2 = sq1 + 1
    inter = np.array([[2, 1], [1, 2]], np.float32)
    allp = np.concatenate([inter, np.zeros((14, 2), np.float32), sq1, sq2])[None]
    mask = np.zeros((1, 24), bool)
    mask[0, [0, 1]] = True          # the two edge intersections
    mask[0, 16 + 2] = True          # (2,2) of sq1 lies inside sq2
    mask[0, 20 + 0] = True          # (1,1) of sq2 lies inside sq1
    idx = convex_sort(_t(allp, cuda), torch.from_numpy(mask).to(cuda)).cpu().numpy()[0]
    idx = np.where(idx == -1, 24, idx)
    ext = np.concatenate([allp[0], np.zeros((1, 2), np.float32)])
    poly = ext[idx]
    area = 0.5 * abs(np.sum(poly[:-1, 0] * poly[1:, 1] - poly[:-1, 1] * poly[1:, 0]))
    assert abs(area - 1.0) < 1e-6
    with pytest.raises(Exception):
        convex_sort(torch.zeros(2, 4, 2), torch.zeros(2, 4))   # CPU tensors: no fallback


# ---------------------------------------------------------------- coder / offsets (a7, a9, a10, a17)
def test_box_coder_roundtrip_and_oracle(cuda):
    from rs_detection_amd import ops
    rng = np.random.default_rng(0)
    prop, gt = dota_boxes(rng, 5000), dota_boxes(rng, 5000)
    stds = (0.1, 0.1, 0.2, 0.2, 0.1)
    want = oracle.np_bbox2delta_rotated(prop, gt, (0,) * 5, stds)
    got = ops.bbox2delta_rotated(_t(prop, cuda), _t(gt, cuda), (0,) * 5, stds)
    # targets reach |1e3| (far-apart random pairs / std 0.1): 1e-4 relative to max(1,|x|)
    assert np.abs((got.cpu().numpy() - want) / np.maximum(np.abs(want), 1)).max() <= TOL
    back = ops.delta2bbox_rotated(_t(prop, cuda), got, (0,) * 5, stds, wh_ratio_clip=1e-6).cpu().numpy()
    wantb = oracle.np_delta2bbox_rotated(prop, want, (0,) * 5, stds, 1e-6)
    assert np.abs(back - wantb).max() <= 2e-3  # coordinates up to 1024 px in fp32
    # decode(encode(gt)) == gt up to angle normalisation
    assert np.abs(back[:, :4] - gt[:, :4]).max() <= 2e-2
    d = rng.standard_normal((5000, 5)).astype(np.float32) * 0.3
    got = ops.delta2bbox_rotated(_t(prop, cuda), _t(d, cuda)).cpu().numpy()
    want = oracle.np_delta2bbox_rotated(prop, d)
    assert np.abs(got - want).max() <= 1e-3 and np.abs((got - want) / np.maximum(np.abs(want), 1)).max() <= TOL


@pytest.mark.parametrize("H,W,stride", [(8, 8, 128), (16, 12, 64), (128, 128, 8)])
def test_refine_and_offset_vs_oracle(cuda, H, W, stride):
    from rs_detection_amd import ops
    rng = np.random.default_rng(H)
    B = 2
    anchors = oracle.np_s2anet_grid_anchors((H, W), stride)
    pred = (rng.standard_normal((B, 5, H, W)) * 0.2).astype(np.float32)
    refined, offset = ops.s2a_refine_and_offset(_t(pred, cuda), _t(anchors, cuda), stride)
    for b in range(B):
        d = pred[b].transpose(1, 2, 0).reshape(-1, 5)
        want_r = oracle.np_delta2bbox_rotated(anchors, d, wh_ratio_clip=1e-6)
        got_r = refined[b].cpu().numpy().reshape(-1, 5)
        assert np.abs((got_r - want_r) / np.maximum(np.abs(want_r), 1)).max() <= TOL
        want_o = oracle.np_align_conv_offset(want_r, (H, W), stride)
        assert np.abs(offset[b].cpu().numpy() - want_o).max() <= 1e-3


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_refine_and_offset_all_levels_in_one_launch(cuda, dt):
    """rsdet_s2a_refine_and_offset_multi (five pyramid levels, one launch; bf16 predictions widened in the kernel) == the
    per-level calls on the fp32 values, bit for bit."""
    from rs_detection_amd.ops.box_coder import s2a_refine_and_offset, s2a_refine_and_offset_levels
    rng = np.random.default_rng(21)
    B, sizes, strides = 3, [(32, 32), (16, 16), (8, 8), (4, 4), (2, 3)], [8, 16, 32, 64, 128]
    anchors = [_t(oracle.np_s2anet_grid_anchors(hw, s), cuda) for hw, s in zip(sizes, strides)]
    preds = [torch.from_numpy((rng.standard_normal((B, 5) + hw) * 0.2).astype(np.float32)).to(cuda).to(dt) for hw in sizes]
    means, stds = (0.1, 0.0, 0.0, 0.05, 0.0), (1.0, 1.0, 0.5, 0.5, 1.0)
    refined, offsets = s2a_refine_and_offset_levels(preds, anchors, strides, 3, means, stds, 1e-6)
    for p, a, s, r, o in zip(preds, anchors, strides, refined, offsets):
        wr, wo = s2a_refine_and_offset(p.float(), a, s, 3, means, stds, 1e-6)
        assert r.dtype == torch.float32 and torch.equal(r, wr) and torch.equal(o, wo)
    only_r, none_o = s2a_refine_and_offset_levels(preds[:2], anchors[:2], strides[:2], 3, means, stds, 1e-6, want_offset=False)
    assert none_o == [None, None] and torch.equal(only_r[1], refined[1])


def test_rotated_box_to_poly(cuda):
    from rs_detection_amd import ops
    b = dota_boxes(np.random.default_rng(4), 1000)
    got = ops.rotated_box_to_poly(_t(b, cuda)).cpu().numpy()
    assert np.abs(got - oracle.np_rotated_box_to_poly(b)).max() <= 1e-3
    assert ops.rotated_box_to_poly(torch.zeros((0, 5), device=cuda)).shape == (0, 8)


def test_ops_refuse_cpu_tensors(cuda):
    """No CPU fallback anywhere in the product path."""
    from rs_detection_amd import ops, _lib
    with pytest.raises(_lib.RsdetError):
        ops.box_iou_rotated(torch.zeros(2, 5), torch.zeros(2, 5))


@pytest.mark.parametrize("shape,relu,with_res,affine", [((4, 64, 56, 56), True, True, True), ((2, 33, 7, 9), True, False, True),
                                                       ((3, 16, 5, 5), False, True, True), ((2, 8, 12, 12), True, True, False),
                                                       ((1, 256, 128, 128), True, True, True)])
def test_bn_act_fused_equals_torch_sequence(cuda, shape, relu, with_res, affine):
    """a21: eval-mode BatchNorm + residual + ReLU in one pass == the torch op sequence (resnet.py:101-126),
    forward and all four gradients; parameter gradients are deterministic (two-stage reduction)."""
    from rs_detection_amd.ops.bn_act import bn_act
    torch.manual_seed(sum(shape))
    N, C, H, W = shape
    bn = torch.nn.BatchNorm2d(C, affine=affine).to(cuda).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 1)
        bn.running_var.uniform_(0.5, 2)
        if affine:
            bn.weight.normal_(1, 0.3)
            bn.bias.normal_(0, 0.3)
    x = torch.randn(shape, device=cuda, requires_grad=True)
    res = torch.randn(shape, device=cuda, requires_grad=True) if with_res else None
    gy = torch.randn(shape, device=cuda)

    def run(fused):
        for t in (x, res) + tuple(bn.parameters()):
            if t is not None:
                t.grad = None
        if fused:
            y = bn_act(x, bn, res, relu)
        else:
            y = bn(x) if res is None else bn(x) + res
            y = torch.relu(y) if relu else y
        y.backward(gy)
        return [y.detach()] + [None if t is None else t.grad.clone() for t in (x, res) + tuple(bn.parameters())]

    want, got, again = run(False), run(True), run(True)
    assert type(got[0].grad_fn).__name__ != "ReluBackward0"
    for a, b, c in zip(want, got, again):
        if a is None:
            assert b is None
            continue
        scale = max(float(a.abs().max()), 1.0)
        assert float((a - b).abs().max()) <= 2e-5 * scale * (10 if a.dim() == 1 else 1), (a.shape, float((a - b).abs().max()))
        assert torch.equal(b, c)  # bitwise repeatable


def test_resnet_bottleneck_uses_fused_path_and_matches_unfused(cuda):
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.models.backbones.resnet import Bottleneck
    import rs_detection_amd.ops.bn_act as B
    torch.manual_seed(0)
    down = torch.nn.Sequential(torch.nn.Conv2d(64, 256, 1, bias=False), torch.nn.BatchNorm2d(256))
    blk = Bottleneck(64, 64, 1, down).to(cuda).eval()   # eval-mode BatchNorm, gradients still flow (norm_eval)
    x = torch.randn(2, 64, 32, 32, device=cuda, requires_grad=True)
    y = blk(x)
    y.sum().backward()
    gx, gw = x.grad.clone(), blk.bn3.weight.grad.clone()
    x.grad = None
    blk.zero_grad()
    orig = B._fusable
    B._fusable = lambda *a: False
    try:
        y2 = blk(x)
        y2.sum().backward()
    finally:
        B._fusable = orig
    assert float((y - y2).abs().max()) <= 1e-4 * float(y2.abs().max())
    assert float((gx - x.grad).abs().max()) <= 1e-4 * float(x.grad.abs().max())
    assert float((gw - blk.bn3.weight.grad).abs().max()) <= 1e-3 * float(blk.bn3.weight.grad.abs().max())


# ---------------------------------------------------------------- depthwise convolution of the VAN backbone (a20)
@pytest.mark.parametrize("N,C,H,W,K,D,bias", [(2, 8, 32, 64, 3, 1, True), (1, 5, 37, 71, 5, 1, True),
                                               (2, 16, 64, 64, 7, 3, True), (1, 3, 9, 7, 7, 3, False),
                                               (2, 64, 100, 132, 3, 1, True), (1, 32, 256, 256, 7, 3, True)])
def test_dwconv_matches_torch_conv2d(cuda, N, C, H, W, K, D, bias):
    """fp32 reference of the same op: torch's conv2d in float64 on the CPU.  Forward 1e-5, gradients 1e-4 relative."""
    from rs_detection_amd.ops.dwconv import DepthwiseConv2d
    torch.manual_seed(K * 100 + C)
    m = DepthwiseConv2d(C, K, padding=D * (K - 1) // 2, dilation=D, bias=bias).to(cuda)
    x = torch.randn(N, C, H, W, device=cuda, requires_grad=True)
    y = m(x)
    go = torch.randn_like(y)
    y.backward(go)
    xd = x.detach().double().cpu().requires_grad_(True)
    wd = m.weight.detach().double().cpu().requires_grad_(True)
    bd = m.bias.detach().double().cpu().requires_grad_(True) if bias else None
    yd = torch.nn.functional.conv2d(xd, wd, bd, 1, D * (K - 1) // 2, D, C)
    yd.backward(go.double().cpu())
    assert float((y.detach().double().cpu() - yd).abs().max()) <= 1e-5 * max(1.0, float(yd.abs().max()))
    for got, want in ((x.grad, xd.grad), (m.weight.grad, wd.grad)) + (((m.bias.grad, bd.grad),) if bias else ()):
        assert float((got.double().cpu() - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))
    # in_bias: conv(x + b_in) with the zero padding left at zero; its gradient = sum of grad_x per channel
    b_in = torch.randn(C, device=cuda, requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    for p in m.parameters():
        p.grad = None
    y2 = m(x2, b_in)
    y2.backward(go)
    xd2 = x.detach().double().cpu().requires_grad_(True)
    bind = b_in.detach().double().cpu().requires_grad_(True)
    wd2 = m.weight.detach().double().cpu().requires_grad_(True)
    yd2 = torch.nn.functional.conv2d(xd2 + bind[None, :, None, None], wd2, bd.detach() if bias else None, 1,
                                     D * (K - 1) // 2, D, C)
    yd2.backward(go.double().cpu())
    assert float((y2.detach().double().cpu() - yd2).abs().max()) <= 1e-5 * max(1.0, float(yd2.abs().max()))
    for got, want in ((x2.grad, xd2.grad), (b_in.grad, bind.grad), (m.weight.grad, wd2.grad)):
        assert float((got.double().cpu() - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))


def test_dwconv_module_is_a_conv2d_and_falls_back_to_torch_where_not_covered(cuda):
    from rs_detection_amd.ops.dwconv import DepthwiseConv2d, dwconv2d
    m = DepthwiseConv2d(6, 3, padding=1)
    assert isinstance(m, torch.nn.Conv2d) and m.groups == 6 and tuple(m.weight.shape) == (6, 1, 3, 3)
    assert sorted(m.state_dict().keys()) == ["bias", "weight"]          # checkpoint keys of nn.Conv2d
    x = torch.randn(1, 6, 8, 8)
    assert torch.allclose(m(x), torch.nn.functional.conv2d(x, m.weight, m.bias, 1, 1, 1, 6))   # CPU: torch's conv
    s2 = DepthwiseConv2d(6, 3, padding=1, stride=2).to(cuda)                                    # stride 2: not covered
    xs = torch.randn(1, 6, 8, 8, device=cuda)
    assert tuple(s2(xs).shape) == (1, 6, 4, 4)
    with pytest.raises(Exception):
        dwconv2d(xs, torch.randn(6, 1, 9, 9, device=cuda), None, 1)      # 9x9: the C ABI says RSDET_EINVAL, loudly
    # bf16 autocast: inputs are cast to fp32 for the stencil
    mm = DepthwiseConv2d(6, 5, padding=2).to(cuda)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = mm(xs)
    assert out.dtype == torch.float32 and torch.isfinite(out).all()


def test_scale_residual_matches_torch(cuda):
    """Layer scale + residual of the VAN block through the fused affine kernels: values and all three gradients."""
    from rs_detection_amd.ops.bn_act import scale_residual
    torch.manual_seed(0)
    x = torch.randn(2, 24, 33, 17, device=cuda, requires_grad=True)
    f = torch.randn(2, 24, 33, 17, device=cuda, requires_grad=True)
    s = (1e-2 * torch.randn(24, device=cuda)).requires_grad_(True)
    y = scale_residual(x, f, s)
    ref = x + s[:, None, None] * f
    assert torch.equal(y, ref)                      # same two roundings: f * s, then + x
    go = torch.randn_like(y)
    g = torch.autograd.grad(y, (x, f, s), go)
    gr = torch.autograd.grad(ref, (x, f, s), go)
    assert torch.equal(g[0], gr[0]) and torch.equal(g[1], gr[1])
    assert float((g[2] - gr[2]).abs().max()) <= 1e-4 * float(gr[2].abs().max())
    xc, fc, sc = x.detach().cpu(), f.detach().cpu(), s.detach().cpu()
    assert torch.equal(scale_residual(xc, fc, sc), xc + sc[:, None, None] * fc)   # CPU tensors: the torch expression


@pytest.mark.parametrize("relu,res", [(True, True), (True, False), (False, True)])
def test_bn_act_bf16_matches_fp32_math_on_the_same_inputs(cuda, relu, res):
    """bf16 activations (the autocast step): the fused pass computes in fp32 and rounds once, so it must sit within
    one bf16 ulp of the fp32 formula evaluated on the same bf16 inputs; parameter gradients are fp32 sums."""
    from rs_detection_amd.ops.bn_act import bn_act, _fusable
    torch.manual_seed(3)
    N, C, H, W = 2, 24, 20, 36
    bn = torch.nn.BatchNorm2d(C).to(cuda).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.5)
        bn.running_var.uniform_(0.5, 2.0)
        bn.weight.normal_(1, 0.2)
        bn.bias.normal_(0, 0.2)
    x = torch.randn(N, C, H, W, device=cuda).bfloat16().requires_grad_(True)
    r = torch.randn(N, C, H, W, device=cuda).bfloat16().requires_grad_(True) if res else None
    assert _fusable(x, bn, r)
    y = bn_act(x, bn, r, relu)
    assert y.dtype == torch.bfloat16
    xf = x.detach().float().requires_grad_(True)
    rf = r.detach().float().requires_grad_(True) if res else None
    w, b = bn.weight.detach().clone().requires_grad_(True), bn.bias.detach().clone().requires_grad_(True)
    yf = ((xf - bn.running_mean[None, :, None, None]) / torch.sqrt(bn.running_var + bn.eps)[None, :, None, None]) \
        * w[None, :, None, None] + b[None, :, None, None]
    if res:
        yf = yf + rf
    if relu:
        yf = torch.relu(yf)
    assert float((y.float() - yf).abs().max()) <= 2 ** -7 * max(1.0, float(yf.abs().max()))
    go = torch.randn_like(yf).bfloat16()
    y.backward(go)
    # reference gradient: the ReLU mask from the bf16 output the kernel really produced
    yf2 = yf if not relu else yf * (y.float() > 0)
    yf2.backward(go.float())
    assert float((x.grad.float() - xf.grad).abs().max()) <= 2 ** -7 * max(1.0, float(xf.grad.abs().max()))
    if res:
        assert float((r.grad.float() - rf.grad).abs().max()) <= 2 ** -7 * max(1.0, float(rf.grad.abs().max()))
    assert bn.weight.grad.dtype == torch.float32
    assert float((bn.weight.grad - w.grad).abs().max()) <= 2e-2 * max(1.0, float(w.grad.abs().max()))
    assert float((bn.bias.grad - b.grad).abs().max()) <= 2e-2 * max(1.0, float(b.grad.abs().max()))


def test_conv_module_fused_bias_relu_matches_unfused(cuda):
    """ConvModule(conv + bias -> ReLU) through bias_act == the generic conv / ReLU sequence: values and all gradients."""
    from rs_detection_amd.models.utils.modules import ConvModule
    torch.manual_seed(1)
    m = ConvModule(16, 24, 3, stride=1, padding=1).to(cuda)
    with torch.no_grad():
        m.conv.bias.normal_(0, 0.5)
    x = torch.randn(2, 16, 20, 28, device=cuda, requires_grad=True)
    assert m._fused_bias_relu(x, True)
    y = m(x)
    ref = torch.relu(torch.nn.functional.conv2d(x, m.conv.weight, m.conv.bias, 1, 1))
    assert float((y - ref).abs().max()) <= 1e-5
    go = torch.randn_like(y)
    g = torch.autograd.grad(y, (x, m.conv.weight, m.conv.bias), go)
    gr = torch.autograd.grad(ref, (x, m.conv.weight, m.conv.bias), go)
    for a, b in zip(g, gr):
        assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max()))
    # no activation requested / CPU tensors: the generic loop
    assert not m._fused_bias_relu(x, False) and not m._fused_bias_relu(x.detach().cpu(), True)
    mc = ConvModule(16, 24, 3, padding=1)
    xc = torch.randn(1, 16, 8, 8)
    assert torch.allclose(mc(xc), torch.relu(torch.nn.functional.conv2d(xc, mc.conv.weight, mc.conv.bias, 1, 1)))


def test_conv_module_fused_bias_relu_under_bf16_autocast_channels_last(cuda):
    """The bf16 line: a channels_last input under bf16 autocast takes the fused bias + ReLU pass too (an NCHW one keeps
    the torch path).  The output tracks the fp32 module within bf16 accuracy; the gradients are compared with the
    unfused module under the same autocast (against fp32 the weight gradient of torch's own bf16 path is 7 % off here)."""
    from rs_detection_amd.models.utils import modules
    from rs_detection_amd.models.utils.modules import ConvModule
    from rs_detection_amd import _lib
    torch.manual_seed(2)
    m = ConvModule(32, 64, 3, stride=1, padding=1).to(cuda)
    assert _lib.load().rsdet_bn_act_nhwc_supported(64)
    with torch.no_grad():
        m.conv.bias.normal_(0, 0.5)
    x = torch.randn(2, 32, 20, 28, device=cuda)
    xcl = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ref = torch.relu(torch.nn.functional.conv2d(x, m.conv.weight, m.conv.bias, 1, 1))
    go = torch.randn_like(ref).bfloat16()
    res = {}
    saved = modules._FUSE_BIAS_RELU_AMP
    try:
        for fused in (True, False):
            modules._FUSE_BIAS_RELU_AMP = fused
            with torch.autocast("cuda", dtype=torch.bfloat16):
                assert m._fused_bias_relu(xcl, True) == fused and not m._fused_bias_relu(x, True)
                y = m(xcl)
            assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
            assert float((y.detach().float() - ref).abs().max()) <= 3e-2 * float(ref.abs().max())
            res[fused] = (y.detach().float(),) + torch.autograd.grad(y, (xcl, m.conv.weight, m.conv.bias), go)
    finally:
        modules._FUSE_BIAS_RELU_AMP = saved
    # a handful of outputs within one bf16 step of zero change their ReLU mask, each moving single weight-gradient
    # elements by |go * x|: compare in the mean, bound the maximum loosely
    for a, b in zip(res[True], res[False]):
        d = (a.float() - b.float()).abs()
        assert float(d.mean()) <= 2e-2 * float(b.float().abs().mean())
        assert float(d.max()) <= 0.2 * float(b.float().abs().max())
    assert res[True][2].dtype == torch.float32 and res[True][3].dtype == torch.float32


@pytest.mark.parametrize("shape", [(2, 128, 64), (1, 100, 49), (3, 7, 260), (512, 256, 49), (2, 4096, 256), (1, 1, 1)])
def test_transpose_last2_equals_torch(cuda, shape):
    """csrc/layout.hip: (B, R, C) -> (B, C, R) bit for bit, full and ragged tiles; the NCHW <-> NHWC helpers on top."""
    from rs_detection_amd.ops.layout import transpose_last2, nchw_to_nhwc, nhwc_to_nchw
    torch.manual_seed(0)
    x = torch.randn(*shape, device=cuda)
    assert torch.equal(transpose_last2(x), x.transpose(1, 2).contiguous())
    y = torch.randn(2, 24, 9, 13, device=cuda)
    n = nchw_to_nhwc(y)
    assert n.is_contiguous() and torch.equal(n, y.permute(0, 2, 3, 1)) and torch.equal(nhwc_to_nchw(n), y)
    assert torch.equal(nchw_to_nhwc(y.double()), y.double().permute(0, 2, 3, 1).contiguous())   # other dtypes: torch


def test_colsum_and_conv2d_bias_channels_last(cuda):
    """rsdet_colsum_* == the fp32 column sum (to summation order), incl. ragged row counts and C = 1 / 5 / 15 / 64;
    conv2d_bias under bf16 autocast on a channels_last input == the module itself (output, input / weight / bias
    gradients) for a 5-channel and a 64-channel convolution, and is the module itself outside that case."""
    from rs_detection_amd import _lib
    from rs_detection_amd.ops.bn_act import conv2d_bias
    lib = _lib.load()
    torch.manual_seed(4)
    for rows, C in ((1, 5), (777, 15), (70000, 5), (4096, 64), (333, 1)):
        x = torch.randn(rows, C, device=cuda)
        for xt, name in ((x, "rsdet_colsum_f32"), (x.bfloat16(), "rsdet_colsum_bf16")):
            out = torch.empty(C, device=cuda)
            wsb = lib.rsdet_colsum_ws_size(rows, C)
            ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=cuda)
            assert getattr(lib, name)(_lib.ptr(xt), rows, C, _lib.ptr(out), _lib.ptr(ws), wsb, _lib.stream_ptr()) == 0
            ref = xt.double().sum(0)
            assert float((out.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max())) * rows ** 0.5
    out = torch.empty(65, device=cuda)
    assert lib.rsdet_colsum_f32(_lib.ptr(x), 10, 65, _lib.ptr(out), None, 0, _lib.stream_ptr()) != 0
    for cout in (5, 64):
        conv = torch.nn.Conv2d(32, cout, 3, padding=1).to(cuda)
        x = torch.randn(2, 32, 24, 40, device=cuda).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        go = torch.randn(2, cout, 24, 40, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
        res = []
        with torch.autocast("cuda", dtype=torch.bfloat16):
            for f in (lambda: conv2d_bias(conv, x), lambda: conv(x)):
                y = f()
                res.append((y.detach().float(),) + tuple(t.float() for t in
                                                          torch.autograd.grad(y, (x, conv.weight, conv.bias), go)))
        for a, b in zip(*res):
            assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max())
    xn = torch.randn(2, 32, 8, 8, device=cuda)
    assert torch.equal(conv2d_bias(conv, xn), conv(xn))        # NCHW fp32: the module itself


def test_deform_conv_bf16_autocast_path_tracks_fp32(cuda):
    """Under bf16 autocast AlignConv's columns are bf16 and its three products run on bf16 MFMA: output and both
    gradients stay within bf16 accuracy of the fp32 path (relative to the largest value), and the output is bf16."""
    from rs_detection_amd.ops import dcn_v1
    from rs_detection_amd.ops.dcn_v1 import DeformConv
    saved, dcn_v1._LOWP_ALIGNCONV = dcn_v1._LOWP_ALIGNCONV, True     # the default
    torch.manual_seed(5)
    B, C, O, H, W = 2, 32, 48, 24, 40
    m = DeformConv(C, O, 3, padding=1).to(cuda)
    x = torch.randn(B, C, H, W, device=cuda, requires_grad=True)
    off = (torch.randn(B, 18, H, W, device=cuda) * 1.5)
    y32 = m(x, off)
    go = torch.randn_like(y32)
    gx32, gw32 = torch.autograd.grad(y32, (x, m.weight), go)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y16 = m(x, off)
    assert y16.dtype == torch.bfloat16
    gx16, gw16 = torch.autograd.grad(y16, (x, m.weight), go.bfloat16())
    assert gx16.dtype == torch.float32 and gw16.dtype == torch.float32
    dcn_v1._LOWP_ALIGNCONV = saved
    for a, b in ((y16.float(), y32), (gx16, gx32), (gw16, gw32)):
        assert float((a - b).abs().max()) <= 3e-2 * float(b.abs().max())


@pytest.mark.parametrize("shape", [(2, 64, 96, 24, 40), (1, 128, 256, 9, 13), (3, 256, 64, 16, 16)])
def test_alignconv_mfma_implicit_gemm_tracks_fp32(cuda, shape):
    """Under bf16 autocast a covered AlignConv geometry runs as ONE implicit-GEMM launch on the bf16 matrix cores
    (csrc/alignconv_mfma.hip), consuming channels_last bf16 activations as they are: output, input gradient and weight
    gradient stay within bf16 accuracy of the fp32 path on the same bf16-valued operands; partial tiles (9x13), several
    images, O != 256; the sampled columns equal the fp32 columns rounded to bf16."""
    from rs_detection_amd.ops import dcn_v1
    from rs_detection_amd.ops.dcn_v1 import DeformConv
    from rs_detection_amd import _lib
    B, C, O, H, W = shape
    torch.manual_seed(11)
    m = DeformConv(C, O, 3, padding=1).to(cuda)
    with torch.no_grad():
        m.weight.copy_(m.weight.bfloat16().float())
    x = torch.randn(B, C, H, W, device=cuda).bfloat16().float().requires_grad_(True)
    off = torch.randn(B, 18, H, W, device=cuda) * 2.0
    off[0, :, 0, 0] = 50.0          # samples far outside the map: zeros
    off[0, 0, 1, 1] = -1.5          # a sample straddling the upper border
    y32 = m(x, off)
    go = torch.randn_like(y32).bfloat16().float()
    gx32, gw32 = torch.autograd.grad(y32, (x, m.weight), go)
    assert dcn_v1._MFMA_ALIGNCONV and dcn_v1._LOWP_ALIGNCONV
    xcl = x.detach().bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert dcn_v1._mfma_geom(xcl, m.weight, 1, 1, 1, 1) is not None
        y16 = m(xcl, off)
    assert y16.dtype == torch.bfloat16 and tuple(y16.shape) == tuple(y32.shape)
    assert y16.is_contiguous(memory_format=torch.channels_last)
    gx16, gw16 = torch.autograd.grad(y16, (xcl, m.weight), go.bfloat16().contiguous(memory_format=torch.channels_last))
    assert gx16.dtype == torch.bfloat16 and gw16.dtype == torch.float32 and gw16.shape == m.weight.shape
    for a, b in ((y16.float(), y32), (gx16.float(), gx32), (gw16, gw32)):
        assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max())
    # the columns the forward keeps == the fp32 columns, rounded
    lib = _lib.load()
    g = _lib.DcnGeom(C, H, W, 3, 3, 1, 1, 1, 1, 1, 1, B, 1)
    out = torch.empty((B, H, W, O), dtype=torch.bfloat16, device=cuda)
    colT = torch.empty((B * H * W, 9 * C), dtype=torch.bfloat16, device=cuda)
    w_flat = m.weight.detach().permute(0, 2, 3, 1).reshape(O, 9 * C).bfloat16()
    x_nhwc = x.detach().permute(0, 2, 3, 1).contiguous().bfloat16()
    assert lib.rsdet_alignconv_fwd_mfma_bf16(_lib.ptr(x_nhwc), _lib.ptr(off), _lib.ptr(w_flat), g, O, 1, _lib.ptr(out),
                                             _lib.ptr(colT), _lib.stream_ptr()) == 0
    cref = dcn_v1.deformable_im2col(x.detach(), off, (3, 3), (1, 1), (1, 1), (1, 1), 1)
    cref = cref.view(C, 9, -1).permute(2, 1, 0).reshape(-1, 9 * C)
    assert float((colT.float() - cref).abs().max()) <= 2 ** -8 * float(cref.abs().max()) + 1e-6
    # NCHW output form == NHWC output form
    out2 = torch.empty((B, O, H, W), dtype=torch.bfloat16, device=cuda)
    assert lib.rsdet_alignconv_fwd_mfma_bf16(_lib.ptr(x_nhwc), _lib.ptr(off), _lib.ptr(w_flat), g, O, 0, _lib.ptr(out2),
                                             None, _lib.stream_ptr()) == 0
    assert torch.equal(out2, out.permute(0, 3, 1, 2))
    # uncovered geometries are refused, not mis-run
    assert not lib.rsdet_alignconv_mfma_supported(_lib.DcnGeom(32, H, W, 3, 3, 1, 1, 1, 1, 1, 1, B, 1), O)
    assert not lib.rsdet_alignconv_mfma_supported(_lib.DcnGeom(C, H, W, 3, 3, 1, 1, 2, 2, 1, 1, B, 1), O)
    assert lib.rsdet_alignconv_fwd_mfma_bf16(_lib.ptr(x_nhwc), _lib.ptr(off), _lib.ptr(w_flat),
                                             _lib.DcnGeom(C, H, W, 3, 3, 1, 1, 1, 1, 1, 1, B, 2), O, 1, _lib.ptr(out),
                                             None, _lib.stream_ptr()) != 0


def test_alignconv_mfma_levels_share_index_and_weight_cast(cuda):
    """Two AlignConv levels under ``shared_gather_index()`` in a bf16 autocast step with a channels_last weight: one index
    build, ONE bf16 cast of the weight for both levels, the input gradient stored as bf16 by the gather, the weight
    gradient handed over in the weight's own (channels_last) layout -- same values as the two calls on their own."""
    from rs_detection_amd.ops import dcn_v1
    from rs_detection_amd.ops.dcn_v1 import DeformConv
    torch.manual_seed(13)
    m = DeformConv(64, 64, 3, padding=1).to(cuda).to(memory_format=torch.channels_last)
    assert not m.weight.is_contiguous() and m.weight.is_contiguous(memory_format=torch.channels_last)
    xs = [torch.randn(2, 64, s, s, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
          for s in (32, 16)]
    offs = [torch.randn(2, 18, s, s, device=cuda) * 2.0 for s in (32, 16)]
    gos = [torch.randn(2, 64, s, s, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last) for s in (32, 16)]
    res = []
    for shared in (False, True):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if shared:
                with dcn_v1.shared_gather_index() as plan:
                    ys = [m(x, o) for x, o in zip(xs, offs)]
                assert len(plan.calls) == 2 and len(plan.operands) == 1
            else:
                ys = [m(x, o) for x, o in zip(xs, offs)]
        g = torch.autograd.grad(ys, xs + [m.weight], gos)
        assert g[0].dtype == torch.bfloat16 and g[2].dtype == torch.float32 and g[2].shape == m.weight.shape
        if shared:
            assert list(plan.built) == [0]
            assert g[2].is_contiguous(memory_format=torch.channels_last)
        res.append([ys[0].float(), ys[1].float()] + [t.float() for t in g])
    for a, b in zip(*res):
        assert float((a - b).abs().max()) <= 1e-2 * float(b.abs().max())


def test_alignconv_mfma_matches_the_oracle(cuda):
    """The implicit-GEMM AlignConv against the CPU oracle (oracle.deform_im2col, dcn_v1.py:132-184): fp32 columns bit
    for bit, fp32 output == oracle columns x weights (float64 product) within 1e-4, bf16 output within bf16 accuracy;
    offsets that leave the map, ragged tiles, several images."""
    from rs_detection_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    B, C, O, H, W = 2, 64, 96, 11, 21
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    off = (rng.standard_normal((B, 18, H, W)) * 2.5).astype(np.float32)
    off[0, :, 0, :3] = 40.0
    wgt = (rng.standard_normal((O, C, 3, 3)) / 24).astype(np.float32)
    geom = dict(kh=3, kw=3, ph=1, pw=1, sh=1, sw=1, dh=1, dw=1)
    col = oracle.c().deform_im2col(x, off, geom).reshape(C, 9, B * H * W)
    want_colT = col.transpose(2, 1, 0).reshape(B * H * W, 9 * C)
    want = (wgt.reshape(O, C * 9).astype(np.float64) @ col.reshape(C * 9, -1).astype(np.float64))
    want = want.reshape(O, B, H, W).transpose(1, 0, 2, 3)
    g = _lib.DcnGeom(C, H, W, 3, 3, 1, 1, 1, 1, 1, 1, B, 1)
    xd = torch.from_numpy(x).to(cuda).permute(0, 2, 3, 1).contiguous()
    wd = torch.from_numpy(wgt).to(cuda).permute(0, 2, 3, 1).reshape(O, 9 * C).contiguous()
    offd = torch.from_numpy(off).to(cuda)
    out = torch.empty((B, O, H, W), device=cuda)
    colT = torch.empty((B * H * W, 9 * C), device=cuda)
    assert lib.rsdet_alignconv_fwd_mfma_f32(_lib.ptr(xd), _lib.ptr(offd), _lib.ptr(wd), g, O, 0, _lib.ptr(out),
                                            _lib.ptr(colT), _lib.stream_ptr()) == 0
    assert (colT.cpu().numpy() == want_colT).all()
    assert np.abs(out.cpu().numpy() - want).max() <= 1e-4
    outb = torch.empty((B, O, H, W), dtype=torch.bfloat16, device=cuda)
    assert lib.rsdet_alignconv_fwd_mfma_bf16(_lib.ptr(xd.bfloat16()), _lib.ptr(offd), _lib.ptr(wd.bfloat16()), g, O, 0,
                                             _lib.ptr(outb), None, _lib.stream_ptr()) == 0
    assert np.abs(outb.float().cpu().numpy() - want).max() <= 2e-2 * np.abs(want).max()


def test_alignconv_mfma_full_size_properties(cuda):
    """At the step's own size (pyramid level 0 of a 4-tile batch: 4 x 256 x 128 x 128, O = 256) through size-independent
    properties: zero offsets == the plain 3 x 3 convolution (fp32 within 1e-4, bf16 within bf16 accuracy); integer
    offsets == the convolution of the shifted image; linearity in the weights; a constant image with in-map samples ->
    every interior column equals the constant."""
    from rs_detection_amd import _lib
    lib = _lib.load()
    torch.manual_seed(21)
    B, C, O, H, W = 4, 256, 256, 128, 128
    g = _lib.DcnGeom(C, H, W, 3, 3, 1, 1, 1, 1, 1, 1, B, 1)
    x = torch.randn(B, C, H, W, device=cuda)
    w = torch.randn(O, C, 3, 3, device=cuda) / 48
    xn = x.permute(0, 2, 3, 1).contiguous()
    wf = w.permute(0, 2, 3, 1).reshape(O, 9 * C).contiguous()
    zero = torch.zeros(B, 18, H, W, device=cuda)

    def run32(xn_, off_, wf_):
        out = torch.empty((B, O, H, W), device=cuda)
        assert lib.rsdet_alignconv_fwd_mfma_f32(_lib.ptr(xn_), _lib.ptr(off_), _lib.ptr(wf_), g, O, 0, _lib.ptr(out), None,
                                                _lib.stream_ptr()) == 0
        return out

    ref = torch.nn.functional.conv2d(x, w, None, 1, 1)
    scale = float(ref.abs().max())
    y0 = run32(xn, zero, wf)
    assert float((y0 - ref).abs().max()) <= 1e-4 * scale
    # bf16 form on the same (bf16-valued) operands
    xb, wb = xn.bfloat16(), wf.bfloat16()
    outb = torch.empty((B, H, W, O), dtype=torch.bfloat16, device=cuda)
    assert lib.rsdet_alignconv_fwd_mfma_bf16(_lib.ptr(xb), _lib.ptr(zero), _lib.ptr(wb), g, O, 1, _lib.ptr(outb), None,
                                             _lib.stream_ptr()) == 0
    refb = torch.nn.functional.conv2d(xb.float().permute(0, 3, 1, 2), wb.float().view(O, 3, 3, C).permute(0, 3, 1, 2),
                                      None, 1, 1)
    assert float((outb.float().permute(0, 3, 1, 2) - refb).abs().max()) <= 1.5e-2 * float(refb.abs().max())
    # every tap moved by (+1 row, -2 columns): the convolution of the image shifted the same way -- away from the border
    # (at the border the shifted image's zero padding and the samples that come into view differ by construction)
    sh = zero.clone()
    sh[:, 0::2] = 1.0
    sh[:, 1::2] = -2.0
    xs = torch.zeros_like(x)
    xs[:, :, :H - 1, 2:] = x[:, :, 1:, :W - 2]
    d = (run32(xn, sh, wf) - torch.nn.functional.conv2d(xs, w, None, 1, 1))[:, :, 3:H - 3, 4:W - 4]
    assert float(d.abs().max()) <= 1e-4 * scale
    # linearity in the weights (same samples): f(2 w1 - 3 w2) == 2 f(w1) - 3 f(w2)
    off = torch.randn(B, 18, H, W, device=cuda) * 2
    w2 = torch.randn_like(wf) / 48
    lhs = run32(xn, off, (2 * wf - 3 * w2).contiguous())
    rhs = 2 * run32(xn, off, wf) - 3 * run32(xn, off, w2)
    assert float((lhs - rhs).abs().max()) <= 2e-4 * float(rhs.abs().max())
    # a constant image: a sample whose four corners are inside the map reproduces the constant
    colT = torch.empty((B * H * W, 9 * C), device=cuda)
    out = torch.empty((B, O, H, W), device=cuda)
    ones = torch.full_like(xn, 0.75)
    small = (torch.rand(B, 18, H, W, device=cuda) - 0.5) * 0.9
    assert lib.rsdet_alignconv_fwd_mfma_f32(_lib.ptr(ones), _lib.ptr(small), _lib.ptr(wf), g, O, 0, _lib.ptr(out),
                                            _lib.ptr(colT), _lib.stream_ptr()) == 0
    inner = colT.view(B, H, W, 9 * C)[:, 2:H - 2, 2:W - 2]
    assert float((inner - 0.75).abs().max()) <= 1e-6


def test_alignconv_mfma_fp32_implicit_gemm_equals_im2col_path(cuda):
    """fp32: a level large enough to fill the chip runs as the exact-fp32 implicit GEMM (v_mfma_f32_32x32x2_f32); the
    output and both gradients equal the im2col + rocBLAS path to fp32 summation-order noise, and the saved columns are
    bit-identical to the im2col kernel's."""
    from rs_detection_amd.ops import dcn_v1
    from rs_detection_amd.ops.dcn_v1 import DeformConv
    from rs_detection_amd import _lib
    torch.manual_seed(3)
    B, C, O, H, W = 2, 32, 64, 128, 192          # 2 x 16 x 12 = 384 position tiles
    m = DeformConv(C, O, 3, padding=1).to(cuda)
    x = torch.randn(B, C, H, W, device=cuda, requires_grad=True)
    off = torch.randn(B, 18, H, W, device=cuda) * 2.0
    go = torch.randn(B, O, H, W, device=cuda)
    res = {}
    saved = dcn_v1._MFMA_ALIGNCONV
    try:
        for on in (True, False):
            dcn_v1._MFMA_ALIGNCONV = on
            assert (dcn_v1._mfma_geom_f32(x, m.weight, (1, 1), (1, 1), (1, 1), 1) is not None)
            y = m(x, off)
            res[on] = (y.detach(),) + torch.autograd.grad(y, (x, m.weight), go)
    finally:
        dcn_v1._MFMA_ALIGNCONV = saved
    assert float((res[True][0] - res[False][0]).abs().max()) <= 1e-5 * float(res[False][0].abs().max())
    # same kernels, but the gather sums its entries in the order an atomic counter handed them out
    assert float((res[True][1] - res[False][1]).abs().max()) <= 1e-5 * float(res[False][1].abs().max())
    assert float((res[True][2] - res[False][2]).abs().max()) <= 1e-4 * float(res[False][2].abs().max())
    # a small level keeps the im2col path
    assert dcn_v1._mfma_geom_f32(x[:, :, :16, :16], m.weight, (1, 1), (1, 1), (1, 1), 1) is None
    lib = _lib.load()
    g = _lib.DcnGeom(C, 20, 31, 3, 3, 1, 1, 1, 1, 1, 1, B, 1)
    xs, offs = x.detach()[:, :, :20, :31].contiguous(), off[:, :, :20, :31].contiguous()
    out = torch.empty((B, O, 20, 31), device=cuda)
    colT = torch.empty((B * 20 * 31, 9 * C), device=cuda)
    w_t = m.weight.detach().permute(0, 2, 3, 1).reshape(O, 9 * C).contiguous()
    assert lib.rsdet_alignconv_fwd_mfma_f32(_lib.ptr(xs.permute(0, 2, 3, 1).contiguous()), _lib.ptr(offs), _lib.ptr(w_t),
                                            g, O, 0, _lib.ptr(out), _lib.ptr(colT), _lib.stream_ptr()) == 0
    cref = dcn_v1.deformable_im2col(xs, offs, (3, 3), (1, 1), (1, 1), (1, 1), 1)
    assert torch.equal(colT, cref.view(C, 9, -1).permute(2, 1, 0).reshape(-1, 9 * C))


def test_f4_ops_empty_and_degenerate_inputs(cuda):
    """Edge cases of the 8(f) rank-4 ops and the depthwise stencil: empty batches / no RoIs return empty results of
    the right shape (and zero gradients), nothing launches on zero elements, wrong geometry fails loudly."""
    from rs_detection_amd.ops.roi_align_rotated import ROIAlignRotated
    from rs_detection_amd.ops.fr import FR, feature_refine
    from rs_detection_amd.ops.dwconv import dwconv2d
    from rs_detection_amd import _lib
    feat = torch.randn(1, 4, 8, 8, device=cuda, requires_grad=True)
    out = ROIAlignRotated((7, 7), 0.5, 2)(feat, torch.zeros(0, 6, device=cuda))
    assert tuple(out.shape) == (0, 4, 7, 7)
    out.sum().backward()
    assert float(feat.grad.abs().sum()) == 0.0
    # FeatureRefine: empty batch; a box far outside the map adds nothing (fr.py:26-28)
    e = feature_refine(torch.zeros(0, 4, 8, 8, device=cuda), torch.zeros(0, 8, 8, 5, device=cuda), 0.125, 1)
    assert tuple(e.shape) == (0, 4, 8, 8)
    f = torch.randn(1, 3, 6, 6, device=cuda)
    far = torch.full((1, 6, 6, 5), 1e6, device=cuda)
    assert torch.equal(FR(1.0, 5)(f, far), f)
    with pytest.raises(_lib.RsdetError):
        feature_refine(f, torch.zeros(1, 5, 5, 5, device=cuda), 1.0, 1)     # boxes do not match the feature map
    # depthwise stencil: empty batch, 1x1 plane (all halo), unsupported kernel size
    w = torch.randn(3, 1, 3, 3, device=cuda)
    assert tuple(dwconv2d(torch.zeros(0, 3, 5, 5, device=cuda), w, None, 1).shape) == (0, 3, 5, 5)
    one = torch.randn(2, 3, 1, 1, device=cuda)
    assert torch.allclose(dwconv2d(one, w, None, 1), one * w[None, :, 0, 1:2, 1:2].reshape(1, 3, 1, 1), atol=1e-6)
    with pytest.raises(_lib.RsdetError):
        dwconv2d(one, torch.randn(3, 1, 3, 3, device=cuda), None, 2)       # (3, dilation 2) is not a covered geometry


# ---- fused focal + smooth-L1 of one S2ANet module (csrc/losses.hip) ---------------------------------------------
@pytest.mark.parametrize("dtype,B,size", [(torch.float32, 2, 256), (torch.float32, 3, 136), (torch.bfloat16, 2, 256)])
def test_s2a_level_losses_vs_oracle_and_torch(cuda, dtype, B, size):
    """(2, L) per-level losses == the oracle's focal / smooth-L1 (focal_loss.py:5-96, smooth_l1_loss.py:5-54) on the
    permuted maps, == the torch modules of the product; gradients == torch autograd of those modules."""
    import oracle
    from rs_detection_amd import ops
    from rs_detection_amd.models.losses.focal_loss import FocalLoss
    from rs_detection_amd.models.losses.smooth_l1_loss import SmoothL1Loss
    rng = np.random.default_rng(B * size)
    C, strides = 15, (8, 16, 32, 64, 128)
    hws = [(-(-size // s)) ** 2 for s in strides]
    A = sum(hws)
    cls = [torch.from_numpy(rng.normal(-2, 2, (B, C, int(h ** 0.5), int(h ** 0.5))).astype(np.float32)).to(cuda).to(dtype)
           for h in hws]
    box = [torch.from_numpy(rng.normal(0, 0.5, (B, 5, int(h ** 0.5), int(h ** 0.5))).astype(np.float32)).to(cuda).to(dtype)
           for h in hws]
    labels = rng.integers(0, C + 1, (B, A)).astype(np.int32)
    labels[rng.random((B, A)) < 0.9] = 0
    lw = (rng.random((B, A)) < 0.95).astype(np.float32)
    bw = np.repeat((labels > 0)[..., None], 5, -1).astype(np.float32)
    bt = (rng.normal(0, 0.5, (B, A, 5)) * bw).astype(np.float32)
    avg = float(max((labels > 0).sum(), 1))
    t = lambda a: torch.from_numpy(a).to(cuda)
    for m in cls + box:
        m.requires_grad_(True)
    got = ops.s2a_level_losses(cls, box, t(labels), t(lw), t(bt), t(bw), torch.tensor(avg, device=cuda), 0.25, 2.0, 1 / 9,
                               1.0, 1.0)
    assert got.shape == (2, 5)
    again = ops.s2a_level_losses(cls, box, t(labels), t(lw), t(bt), t(bw), torch.tensor(avg, device=cuda), 0.25, 2.0,
                                 1 / 9, 1.0, 1.0)
    assert torch.equal(got, again)                       # deterministic reduction order
    focal, sl1 = FocalLoss(True, 2.0, 0.25), SmoothL1Loss(1 / 9)
    want_t, s = [], 0
    for l, h in enumerate(hws):
        cs = cls[l].float().permute(0, 2, 3, 1).reshape(-1, C)
        bp = box[l].float().permute(0, 2, 3, 1).reshape(-1, 5)
        sl = slice(s, s + h)
        a = focal(cs, t(labels)[:, sl].reshape(-1), t(lw)[:, sl].reshape(-1), avg_factor=avg)
        b = sl1(bp, t(bt)[:, sl].reshape(-1, 5), t(bw)[:, sl].reshape(-1, 5), avg_factor=avg)
        want_t.append((a, b))
        # oracle (float64 restatement) on the values the kernel saw
        oa = oracle.np_sigmoid_focal_loss(cs.detach().cpu().numpy(), labels[:, sl].reshape(-1), lw[:, sl].reshape(-1), 2.0,
                                          0.25, avg)
        ob = oracle.np_smooth_l1_loss(bp.detach().cpu().numpy(), bt[:, sl].reshape(-1, 5), bw[:, sl].reshape(-1, 5), 1 / 9,
                                      avg)
        assert abs(float(got[0, l]) - oa) <= 1e-5 * max(abs(oa), 1e-3) + 1e-6, (l, float(got[0, l]), oa)
        assert abs(float(got[1, l]) - ob) <= 1e-5 * max(abs(ob), 1e-3) + 1e-6, (l, float(got[1, l]), ob)
        s += h
    # gradients: random upstream weights per scalar
    up = torch.from_numpy(rng.uniform(0.5, 1.5, (2, 5)).astype(np.float32)).to(cuda)
    g_mine = torch.autograd.grad((got * up).sum(), cls + box)
    ref = sum(a * up[0, l] + b * up[1, l] for l, (a, b) in enumerate(want_t))
    g_ref = torch.autograd.grad(ref, cls + box)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    for gm, gr in zip(g_mine, g_ref):
        assert gm.dtype == gr.dtype == dtype and gm.shape == gr.shape
        scale = float(gr.float().abs().max()) + 1e-12
        assert float((gm.float() - gr.float()).abs().max()) <= tol * scale + 1e-9


def test_rroi_backward_nchw_form_equals_channels_last_form(cuda, monkeypatch):
    """rsdet_rroi_align_v1_backward_gather_nchw_f32 (NCHW written directly) == the default channels-last gather."""
    from rs_detection_amd.ops import roi_align_rotated_v1
    rng = np.random.default_rng(8)
    for (N, C, H, W, R) in ((2, 256, 64, 64, 60), (1, 20, 37, 41, 9), (3, 64, 16, 16, 5), (1, 512, 23, 27, 30), (2, 18, 9, 9, 4)):
        feat = _t(rng.standard_normal((N, C, H, W)).astype(np.float32), cuda)
        rois = _t(_rois(rng, R, N, W * 4), cuda)
        go = _t(rng.standard_normal((R, C, 7, 7)).astype(np.float32), cuda)
        grads = []
        import importlib
        rroi_mod = importlib.import_module("rs_detection_amd.ops.roi_align_rotated_v1")   # (the package attribute is the function)
        monkeypatch.setattr(rroi_mod, "_INDEX_AT_FORWARD", False)       # (the indexed route always writes NCHW)
        for flag in (False, True):
            monkeypatch.setattr(rroi_mod, "_NCHW_GATHER", flag)
            f = feat.clone().requires_grad_(True)
            roi_align_rotated_v1(f, rois, (7, 7), 0.25, 2).backward(go)
            grads.append(f.grad)
        # same entries, same weights; the two kernels pair the additions differently (1-2 ulp)
        assert float((grads[0] - grads[1]).abs().max()) <= 1e-5 * float(grads[0].abs().max())


def test_rroi_backward_index_built_at_forward_time_equals_the_one_call_form(cuda, monkeypatch):
    """The inverted index of the gather-form backward built on a side stream at forward time
    (rsdet_rroi_align_v*_backward_index_f32 + rsdet_rroi_align_backward_gather_indexed_f32) gives the one-call form's
    gradient -- both variants, several calls in flight before any backward runs."""
    import importlib
    from rs_detection_amd.ops import roi_align_rotated_v1
    from rs_detection_amd.ops.roi_align_rotated import roi_align as roi_align_rotated
    rroi_mod = importlib.import_module("rs_detection_amd.ops.roi_align_rotated_v1")
    rng = np.random.default_rng(18)
    cases = []
    for (N, C, H, W, R) in ((2, 256, 64, 64, 60), (1, 20, 37, 41, 9), (2, 64, 128, 128, 300), (1, 512, 23, 27, 30)):
        feat = _t(rng.standard_normal((N, C, H, W)).astype(np.float32), cuda)
        rois = _t(_rois(rng, R, N, W * 4), cuda)
        go = _t(rng.standard_normal((R, C, 7, 7)).astype(np.float32), cuda)
        cases.append((feat, rois, go))
    res = []
    for flag in (True, False):
        monkeypatch.setattr(rroi_mod, "_INDEX_AT_FORWARD", flag)
        grads = []
        for fn in (roi_align_rotated_v1, roi_align_rotated):
            leaves = [c[0].clone().requires_grad_(True) for c in cases]
            outs = [fn(f, c[1], (7, 7), 0.25, 2) for f, c in zip(leaves, cases)]       # all forwards first
            for o, c in zip(outs, cases):
                o.backward(c[2])
            grads += [f.grad for f in leaves]
        res.append(grads)
    for a, b in zip(*res):            # same entries and weights; their order inside a pixel follows the fill's atomics
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,relu,with_res", [((2, 64, 24, 20), True, False), ((3, 256, 9, 7), True, True),
                                                 ((2, 2048, 5, 6), False, True), ((1, 1024, 8, 8), True, True)])
def test_bn_act_channels_last_equals_nchw_form(cuda, dtype, shape, relu, with_res):
    """The NHWC kernels (the bf16 trunk in channels_last) give the NCHW kernels' results: outputs bit for bit (same
    per-element arithmetic), parameter gradients to summation-order round-off; layouts are preserved."""
    from rs_detection_amd.ops.bn_act import bn_act
    torch.manual_seed(shape[1])
    bn = torch.nn.BatchNorm2d(shape[1]).to(cuda).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.5), bn.running_var.uniform_(0.5, 2.0), bn.weight.normal_(1, 0.3), bn.bias.normal_(0, 0.3)
    x = torch.randn(shape, device=cuda).to(dtype)
    r = torch.randn(shape, device=cuda).to(dtype) if with_res else None
    go = torch.randn(shape, device=cuda).to(dtype)
    outs = []
    for cl in (False, True):
        fmt = torch.channels_last if cl else torch.contiguous_format
        xi = x.clone().contiguous(memory_format=fmt).requires_grad_(True)
        ri = r.clone().contiguous(memory_format=fmt).requires_grad_(True) if with_res else None
        bn.zero_grad()
        y = bn_act(xi, bn, ri, relu)
        assert y.is_contiguous(memory_format=fmt)
        y.backward(go.contiguous(memory_format=fmt))
        outs.append((y.detach(), xi.grad, ri.grad if with_res else None, bn.weight.grad.clone(), bn.bias.grad.clone()))
        assert xi.grad.is_contiguous(memory_format=fmt)
    (y0, gx0, gr0, gw0, gb0), (y1, gx1, gr1, gw1, gb1) = outs
    assert torch.equal(y0, y1) and torch.equal(gx0, gx1)
    if with_res:
        assert torch.equal(gr0, gr1)
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert float((gw0 - gw1).abs().max()) <= tol * (float(gw0.abs().max()) + 1.0)
    assert float((gb0 - gb1).abs().max()) <= tol * (float(gb0.abs().max()) + 1.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,with_res", [((2, 64, 24, 20), False), ((3, 256, 9, 7), True), ((2, 12, 5, 6), True),
                                            ((1, 2048, 4, 4), True)])
def test_bn_act_relu_bit_mask_equals_the_y_gate(cuda, monkeypatch, dtype, shape, with_res):
    """Channels-last bn_act with a ReLU: the backward that reads the forward's one-bit-per-element gate gives the very
    gradients of the backward that re-reads y (outputs of exactly 0 and of -0 included: both gate the gradient off);
    shapes with no mask form (bf16, eight-channel kernels, odd pixel count: (3, 256, 9, 7)) quietly keep y."""
    from rs_detection_amd.ops import bn_act as mod
    from rs_detection_amd import _lib
    torch.manual_seed(shape[1] + shape[2])
    C = shape[1]
    bn = torch.nn.BatchNorm2d(C).to(cuda).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.5), bn.running_var.uniform_(0.5, 2.0), bn.weight.normal_(1, 0.3), bn.bias.normal_(0, 0.3)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    x = torch.randn(shape, device=cuda).to(dtype)
    r = torch.randn(shape, device=cuda).to(dtype) if with_res else None
    if with_res:                       # some outputs exactly 0: bn(x) + r == 0 where r = -bn(x) (as stored)
        with torch.no_grad():
            r[:, :, 0, :] = -mod.bn_act(cl(x), bn, None, False)[:, :, 0, :]
    go = cl(torch.randn(shape, device=cuda).to(dtype))
    outs = []
    for use_mask in (True, False):
        monkeypatch.setattr(mod, "_RELU_MASK", use_mask)
        xi = cl(x.clone()).requires_grad_(True)
        ri = cl(r.clone()).requires_grad_(True) if with_res else None
        bn.zero_grad()
        y = mod.bn_act(xi, bn, ri, True)
        if "BNAct" in type(y.grad_fn).__name__:        # (a channel count the NHWC kernels do not tile -- 12 -- runs torch)
            saved = y.grad_fn.saved_tensors[1]
            nb = _lib.load().rsdet_bn_act_relu_mask_bytes(shape[0], C, shape[2] * shape[3], int(dtype == torch.bfloat16))
            assert (saved.dtype == torch.uint8 and saved.numel() == nb) if (use_mask and nb) else saved.dtype == dtype
        y.backward(go)
        outs.append((y.detach().clone(), xi.grad.clone(), ri.grad.clone() if with_res else None, bn.weight.grad.clone(),
                     bn.bias.grad.clone()))
    for a, b in zip(*outs):
        if a is not None:
            assert torch.equal(a, b)
    if with_res and dtype == torch.float32:   # (in bf16 bn(x) is rounded before it is negated: the sums are tiny, not 0)
        assert bool((outs[0][0][:, :, 0, :] == 0).all()) and bool((outs[0][2][:, :, 0, :] == 0).all())


@pytest.mark.parametrize("C,H,W", [(64, 20, 36), (4, 20, 36), (256, 9, 7), (2048, 3, 5), (1024, 4, 4)])
@pytest.mark.parametrize("relu,res", [(True, True), (True, False), (False, True), (False, False)])
def test_bn_act_bf16_channels_last_forms(cuda, C, H, W, relu, res):
    """The channels_last bf16 kernels of the autocast step -- eight channels (16 bytes) per lane where C / 8 divides 256
    (64, 256, 1024, 2048 here), four per lane otherwise (4) -- against the fp32 formula on the same bf16 inputs:
    outputs within one bf16 ulp, input / residual gradients likewise, parameter gradients as fp32 sums."""
    from rs_detection_amd.ops.bn_act import bn_act, _fusable
    torch.manual_seed(C + H)
    N = 3
    bn = torch.nn.BatchNorm2d(C).to(cuda).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.5)
        bn.running_var.uniform_(0.5, 2.0)
        bn.weight.normal_(1, 0.2)
        bn.bias.normal_(0, 0.2)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    x = cl(torch.randn(N, C, H, W, device=cuda).bfloat16()).requires_grad_(True)
    r = cl(torch.randn(N, C, H, W, device=cuda).bfloat16()).requires_grad_(True) if res else None
    assert _fusable(x, bn, r)
    y = bn_act(x, bn, r, relu)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    xf = x.detach().float().requires_grad_(True)
    rf = r.detach().float().requires_grad_(True) if res else None
    w, b = bn.weight.detach().clone().requires_grad_(True), bn.bias.detach().clone().requires_grad_(True)
    yf = ((xf - bn.running_mean[None, :, None, None]) / torch.sqrt(bn.running_var + bn.eps)[None, :, None, None]) \
        * w[None, :, None, None] + b[None, :, None, None]
    if res:
        yf = yf + rf
    if relu:
        yf = torch.relu(yf)
    assert float((y.float() - yf).abs().max()) <= 2 ** -7 * max(1.0, float(yf.abs().max()))
    go = cl(torch.randn_like(yf).bfloat16())
    y.backward(go)
    yf2 = yf if not relu else yf * (y.float() > 0)
    yf2.backward(go.float())
    assert float((x.grad.float() - xf.grad).abs().max()) <= 2 ** -7 * max(1.0, float(xf.grad.abs().max()))
    if res:
        assert float((r.grad.float() - rf.grad).abs().max()) <= 2 ** -7 * max(1.0, float(rf.grad.abs().max()))
    assert bn.weight.grad.dtype == torch.float32
    assert float((bn.weight.grad - w.grad).abs().max()) <= 2e-2 * max(1.0, float(w.grad.abs().max()))
    assert float((bn.bias.grad - b.grad).abs().max()) <= 2e-2 * max(1.0, float(b.grad.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C,H,W", [(2, 64, 40, 56), (1, 64, 33, 47), (3, 16, 7, 9), (1, 8, 1, 1)])
def test_bn_relu_maxpool_stem_tail_is_bit_identical(cuda, dtype, N, C, H, W):
    """The fused stem tail (bn1 -> relu -> maxpool 3x3/s2/p1 in one pass, csrc/bn_act.hip) == maxpool(bn_act(x)) bit for
    bit, odd sizes and borders included; with a gradient recorded it falls back to the unfused sequence."""
    from rs_detection_amd.ops.bn_act import bn_relu_maxpool, bn_act
    torch.manual_seed(N * 100 + H)
    bn = torch.nn.BatchNorm2d(C).to(cuda).eval()
    with torch.no_grad():
        bn.running_mean.normal_(0, 0.5)
        bn.running_var.uniform_(0.5, 2.0)
        bn.weight.normal_(0, 1.0)          # negative scales too: max and the affine do not commute
        bn.bias.normal_(0, 0.5)
    pool = torch.nn.MaxPool2d(3, 2, 1)
    x = torch.randn(N, C, H, W, device=cuda).to(dtype)
    x[0, 0, 0, 0] = float("nan")                 # non-finite inputs: as through bn_act (its ReLU maps NaN to 0) + max_pool2d
    x[-1, -1, -1, -1] = float("inf")
    x = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        got = bn_relu_maxpool(x, bn, pool)
        want = pool(bn_act(x, bn))
    assert got.shape == want.shape == (N, C, (H + 1) // 2, (W + 1) // 2)
    assert got.is_contiguous(memory_format=torch.channels_last)
    assert not torch.isnan(want).any() and torch.equal(got, want)
    xg = x.clone().requires_grad_(True)
    y = bn_relu_maxpool(xg, bn, pool)           # gradient recorded: the unfused path, differentiable
    y.float().sum().backward()
    assert xg.grad is not None and torch.equal(y.detach(), want)
