"""GPU: the CUDA-only ops against the REFERENCE'S OWN DEVICE CODE (oracle/_ref/libjdet_ref_hip.so =
/root/reference/python/jdet/ops/{dcn_v1,roi_align_rotated_v1,fr,nms_poly}.py kernel text compiled unmodified by hipcc
for gfx950: oracle/build_ref_hip.py) at the shapes the BASELINE configs run them at -- S2ANet pyramid level 0
(4 x 256 x 128 x 128) for DCN / FeatureRefine, the Oriented R-CNN head (512 RoIs on 2 x 256 x 256 x 256) for RROIAlign,
2 000 quadrilaterals for polygon NMS.  Rounds 1-4 pinned these ops through a single-threaded HOST shim of the same text
(tests/golden/{dcn,rroi,fr,poly_nms}.npz); this is the harder pin: device sinf/cosf, the compiler's FMA contraction, the
reference's own launch geometry and atomics.  Tolerance 1e-4 (north star); keep lists bit-exact.

Test infrastructure only: nothing under rs_detection_amd/ loads the library (tests/test_abi_cpu.py guards that)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GEOM = dict(kh=3, kw=3, ph=1, pw=1, sh=1, sw=1, dh=1, dw=1)
K3 = ((3, 3), (1, 1), (1, 1), (1, 1))


@pytest.fixture(scope="module")
def rh():
    import oracle
    r = oracle.ref_hip()
    if not r.available:
        pytest.skip("oracle/_ref/libjdet_ref_hip.so not built (needs /root/reference at build time)")
    return r


def _close(a, b, tol=1e-4):
    a, b = a.float(), b.float()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, (err, scale)
    return err


def _alignconv_offsets(B, H, W, dev, seed):
    """Offsets shaped like AlignConv's (s2anet_head.py:657-723): a rotated, scaled 3x3 grid minus the regular one --
    several pixels large, reaching outside the map at the border."""
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, 18, H, W, generator=g) * 2.5).to(dev)


@pytest.mark.parametrize("B,C,H,W", [(4, 256, 128, 128), (2, 64, 37, 53)])
def test_deform_im2col_col2im_against_reference_device_code(cuda, rh, B, C, H, W):
    from rs_detection_amd import ops
    g = torch.Generator().manual_seed(B * C)
    im = torch.randn(B, C, H, W, generator=g).to(cuda)
    off = _alignconv_offsets(B, H, W, cuda, 3)
    want = rh.deform_im2col(im, off, GEOM)                                   # (C*9, B, H, W)
    got = ops.deformable_im2col(im, off, *K3)
    _close(got.reshape(want.shape), want)
    # the channels-last pair the train step uses: same values, transposed layout
    colT = ops.deformable_im2col_nhwc(im.permute(0, 2, 3, 1).contiguous(), off, *K3)       # (B*H*W, 9*C) tap-major
    want_T = want.view(C, 9, B, H * W).permute(2, 3, 1, 0).reshape(B * H * W, 9 * C)
    _close(colT, want_T)
    # backward-data: the reference's atomic scatter vs our scatter and our gather form
    gcol = torch.randn(want.shape, generator=g).to(cuda)
    want_gim = rh.deform_col2im(gcol, off, im.shape, GEOM)
    _close(ops.deformable_col2im(gcol.reshape(got.shape), off, im.shape, *K3), want_gim)
    gcolT = gcol.view(C, 9, B, H * W).permute(2, 3, 1, 0).reshape(B * H * W, 9 * C).contiguous()
    from rs_detection_amd.ops.dcn_v1 import deformable_col2im_gather_nhwc
    gim_nhwc = deformable_col2im_gather_nhwc(gcolT, off, (B, H, W, C), *K3)
    _close(gim_nhwc.permute(0, 3, 1, 2), want_gim)
    # offset gradient (computed by the reference's backward, unused by S2ANet: SURVEY q16)
    want_goff = rh.deform_col2im_coord(gcol, im, off, GEOM)
    _close(ops.deformable_col2im_coord(gcol.reshape(got.shape), im, off, *K3), want_goff)


def test_alignconv_implicit_gemm_columns_against_reference_device_code(cuda, rh):
    """What the S2ANet step launches at level 0 (csrc/alignconv_mfma.hip, fp32): its sampled columns and its product
    against the reference's device im2col + an fp64 GEMM."""
    from rs_detection_amd import _lib
    B, C, O, H, W = 2, 256, 256, 128, 128
    g = torch.Generator().manual_seed(11)
    im = torch.randn(B, C, H, W, generator=g).to(cuda)
    off = _alignconv_offsets(B, H, W, cuda, 5)
    wgt = (torch.randn(O, C, 3, 3, generator=g) / 48).to(cuda)
    want = rh.deform_im2col(im, off, GEOM)                                   # row = c*9 + tap
    xd = im.permute(0, 2, 3, 1).contiguous()
    wd = wgt.permute(0, 2, 3, 1).reshape(O, 9 * C).contiguous()
    outd = torch.empty((B, O, H, W), device=cuda)
    colT = torch.empty((B * H * W, 9 * C), device=cuda)
    lib = _lib.load()
    rc = lib.rsdet_alignconv_fwd_mfma_f32(_lib.ptr(xd), _lib.ptr(off), _lib.ptr(wd),
                                          _lib.DcnGeom(C, H, W, 3, 3, 1, 1, 1, 1, 1, 1, B, 1), O, 0, _lib.ptr(outd),
                                          _lib.ptr(colT), _lib.stream_ptr())
    assert rc == 0
    _close(colT, want.view(C, 9, B, H * W).permute(2, 3, 1, 0).reshape(B * H * W, 9 * C))
    ref_out = (wgt.reshape(O, C * 9).double() @ want.reshape(C * 9, -1).double()).reshape(O, B, H, W).permute(1, 0, 2, 3)
    _close(outd, ref_out.float(), 2e-4)      # fp32 MFMA accumulation over K = 2304


def _rois(n, N, span, seed):
    from conftest import dota_boxes
    rng = np.random.default_rng(seed)
    b = dota_boxes(rng, n, span, 8, 240, 120)
    r = np.concatenate([rng.integers(0, N, (n, 1)).astype(np.float32), b], 1)
    r[0, 1:3] = [-40, -40]           # partially outside the map
    return r


@pytest.mark.parametrize("N,C,H,W,R,scale,sr", [(2, 256, 256, 256, 512, 0.25, 2), (2, 256, 32, 32, 200, 1 / 32., 2),
                                                 (1, 16, 40, 56, 23, 0.125, 0)])
def test_rroi_align_against_reference_device_code(cuda, rh, N, C, H, W, R, scale, sr):
    from rs_detection_amd.ops import roi_align_rotated_v1
    g = torch.Generator().manual_seed(R)
    feat = torch.randn(N, C, H, W, generator=g).to(cuda).requires_grad_(True)
    rois = torch.from_numpy(_rois(R, N, W / scale, R)).to(cuda)
    out = roi_align_rotated_v1(feat, rois, (7, 7), scale, sr)
    _close(out.detach(), rh.rroi_forward(feat.detach(), rois, (7, 7), scale, sr))
    go = torch.randn(out.shape, generator=g).to(cuda)
    out.backward(go)
    _close(feat.grad, rh.rroi_backward(go, rois, feat.shape, scale, sr))


@pytest.mark.parametrize("points", [1, 5])
@pytest.mark.parametrize("N,C,H", [(2, 256, 128), (1, 24, 20)])
def test_feature_refine_against_reference_device_code(cuda, rh, N, C, H, points):
    from rs_detection_amd.ops.fr import feature_refine
    rng = np.random.default_rng(C + points)
    feat = torch.from_numpy(rng.standard_normal((N, C, H, H)).astype(np.float32)).to(cuda).requires_grad_(True)
    yc, xc = np.meshgrid(8.0 * np.arange(H), 8.0 * np.arange(H), indexing="ij")
    bx = np.stack([xc[None] + 32 * rng.standard_normal((N, H, H)), yc[None] + 32 * rng.standard_normal((N, H, H)),
                   32 * np.exp(rng.standard_normal((N, H, H))), 32 * np.exp(rng.standard_normal((N, H, H))),
                   -np.pi / 2 * rng.random((N, H, H))], -1).astype(np.float32)
    bx = torch.from_numpy(bx).to(cuda)
    out = feature_refine(feat, bx, 0.125, points)
    _close(out.detach(), rh.fr_forward(feat.detach(), bx, 0.125, points))
    go = torch.from_numpy(rng.standard_normal(out.shape).astype(np.float32)).to(cuda)
    out.backward(go)
    # the reference's backward kernel adds the identity term itself (`atomicAdd(bottom_diff + index, top_diff[index])`,
    # fr.py:212) onto an output its wrapper means to be zero (`jt.zeros_like`, :245): the launcher zero-fills it
    _close(feat.grad, rh.fr_backward(go, bx, 0.125, points))


@pytest.mark.parametrize("n", [2000, 333])
def test_poly_nms_against_reference_device_code(cuda, rh, n):
    """Polygon NMS (nms_poly.py:135-231).  The reference sums signed fp32 triangle intersections of ~1e5 px^2 each, so
    its own IoUs move by ~3e-3 with the compiler's FMA contraction (measured here: device build with contraction on vs
    off).  rsdet_poly_* follows the UNCONTRACTED arithmetic (the CPU / -fmad=false semantics the golden fixtures pin):
    bit-identical to the reference text compiled with contraction off, within that contraction noise of the default
    build; keep lists identical to the former, and to the latter up to pairs that sit inside the noise of the threshold."""
    from rs_detection_amd.ops import poly_nms, poly_iou_f32
    from rs_detection_amd.ops.box_coder import rotated_box_to_poly
    from rs_detection_amd.utils import synthetic as syn
    d, sc, _ = syn.nms_cluster_boxes(n)
    q = rotated_box_to_poly(torch.from_numpy(d).to(cuda))
    dets = torch.cat([q, torch.from_numpy(sc).to(cuda)[:, None]], 1).contiguous()
    order = torch.argsort(dets[:, 8], descending=True, stable=True)
    srt = dets[order].contiguous()
    m = min(n, 400)
    ours = poly_iou_f32(srt[:m, :8].contiguous(), srt[:m, :8].contiguous())
    ii, jj = torch.meshgrid(torch.arange(m, device=cuda), torch.arange(m, device=cuda), indexing="ij")
    pa, pb = srt[ii.reshape(-1), :8].contiguous(), srt[jj.reshape(-1), :8].contiguous()
    exact = rh.poly_iou_pairs(pa, pb, contract=False).view(m, m)
    assert (ours == exact).all(), float((ours - exact).abs().max())
    fma = rh.poly_iou_pairs(pa, pb, contract=True).view(m, m)
    noise = float((fma - exact).abs().max())
    print("poly IoU: contraction moves the reference by", noise)
    assert noise <= 1e-2
    for thr in (0.1, 0.5):
        got = poly_nms(dets, thr).cpu().numpy()
        want = order.cpu().numpy()[rh.poly_nms_sorted(srt, thr, contract=False)]
        assert len(got) == len(want) and (got == want).all(), (thr, len(got), len(want))
        other = order.cpu().numpy()[rh.poly_nms_sorted(srt, thr, contract=True)]
        diff = len(set(got.tolist()) ^ set(other.tolist()))
        print("poly NMS thr %g: %d kept, %d differ under contraction" % (thr, len(got), diff))
        assert diff <= max(4, n // 50)
