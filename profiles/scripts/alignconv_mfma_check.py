"""AlignConv implicit GEMM (csrc/alignconv_mfma.hip): parity against the fp32 deform_conv on the same bf16-valued
operands, and HIP-graph-free event timing at the S2ANet pyramid shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rs_detection_amd import _lib
from rs_detection_amd.ops import dcn_v1

dev = torch.device("cuda:0")
lib = _lib.load()


def run(x_nhwc, off, w_flat, O, out_nhwc=True, want_col=False):
    B, H, W, C = x_nhwc.shape
    g = _lib.DcnGeom(C, H, W, 3, 3, 1, 1, 1, 1, 1, 1, B, 1)
    out = torch.empty((B, H, W, O) if out_nhwc else (B, O, H, W), dtype=torch.bfloat16, device=dev)
    col = torch.empty((B * H * W, 9 * C), dtype=torch.bfloat16, device=dev) if want_col else None
    rc = lib.rsdet_alignconv_fwd_mfma_bf16(_lib.ptr(x_nhwc), _lib.ptr(off), _lib.ptr(w_flat), g, O, int(out_nhwc),
                                           _lib.ptr(out), _lib.ptr(col), _lib.stream_ptr())
    assert rc == 0, rc
    return out, col


def run32(x_nhwc, off, w_flat, O, out_nhwc=True, want_col=False):
    B, H, W, C = x_nhwc.shape
    g = _lib.DcnGeom(C, H, W, 3, 3, 1, 1, 1, 1, 1, 1, B, 1)
    out = torch.empty((B, H, W, O) if out_nhwc else (B, O, H, W), dtype=torch.float32, device=dev)
    col = torch.empty((B * H * W, 9 * C), dtype=torch.float32, device=dev) if want_col else None
    rc = lib.rsdet_alignconv_fwd_mfma_f32(_lib.ptr(x_nhwc), _lib.ptr(off), _lib.ptr(w_flat), g, O, int(out_nhwc),
                                          _lib.ptr(out), _lib.ptr(col), _lib.stream_ptr())
    assert rc == 0, rc
    return out, col


def check32(B, C, O, H, W, scale):
    torch.manual_seed(0)
    x = torch.randn(B, C, H, W, device=dev)
    wgt = torch.randn(O, C, 3, 3, device=dev) / (3 * C ** 0.5)
    off = torch.randn(B, 18, H, W, device=dev) * scale
    dcn_v1._LOWP_ALIGNCONV = False
    ref = dcn_v1.deform_conv(x, off, wgt, 1, 1, 1, 1, 1)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    w_flat = wgt.permute(0, 2, 3, 1).reshape(O, 9 * C).contiguous()
    out, col = run32(x_nhwc, off, w_flat, O, False, True)
    err = float((out - ref).abs().max()) / float(ref.abs().max())
    cref = dcn_v1.deformable_im2col(x, off, (3, 3), (1, 1), (1, 1), (1, 1), 1)
    cref = cref.view(C, 9, -1).permute(2, 1, 0).reshape(-1, 9 * C)
    print(f"f32 B{B} C{C} O{O} {H}x{W}: out rel err {err:.2e}  col equal {bool(torch.equal(col, cref))}")


def bench32(B, C, O, H, W, want_col):
    x_nhwc = torch.randn(B, H, W, C, device=dev)
    w_flat = torch.randn(O, 9 * C, device=dev) / 48
    off = anchor_offsets(B, H, W)
    for _ in range(3):
        run32(x_nhwc, off, w_flat, O, False, want_col)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        run32(x_nhwc, off, w_flat, O, False, want_col)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    fl = 2.0 * B * H * W * O * 9 * C
    print(f"bench f32 B{B} C{C} O{O} {H}x{W} col={want_col}: {us:.1f} us  {fl / us / 1e6:.1f} TFLOP/s")


def check(B, C, O, H, W, scale):
    torch.manual_seed(0)
    x = torch.randn(B, C, H, W, device=dev).bfloat16().float()
    wgt = (torch.randn(O, C, 3, 3, device=dev) / (3 * C ** 0.5)).bfloat16().float()
    off = torch.randn(B, 18, H, W, device=dev) * scale
    dcn_v1._LOWP_ALIGNCONV = False
    ref = dcn_v1.deform_conv(x, off, wgt, 1, 1, 1, 1, 1)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().bfloat16()
    w_flat = wgt.permute(0, 2, 3, 1).reshape(O, 9 * C).contiguous().bfloat16()
    for nhwc in (True, False):
        out, col = run(x_nhwc, off, w_flat, O, nhwc, True)
        o = out.float().permute(0, 3, 1, 2) if nhwc else out.float()
        err = float((o - ref).abs().max()) / float(ref.abs().max())
        # columns: the fp32 columns of the reference layout, rounded
        cref = dcn_v1.deformable_im2col(x, off, (3, 3), (1, 1), (1, 1), (1, 1), 1)  # (C*9, B*H*W), row = c*9 + t
        cref = cref.view(C, 9, -1).permute(2, 1, 0).reshape(-1, 9 * C)
        cerr = float((col.float() - cref).abs().max())
        print(f"B{B} C{C} O{O} {H}x{W} nhwc={nhwc}: out rel err {err:.2e}  col abs err {cerr:.2e}")


REAL = True


def anchor_offsets(B, H, W):
    """AlignConv offsets of smooth rotated boxes (s2anet_head.py:603-660): the 3x3 taps stretched over a (w, h) box of
    ~4 feature pixels, rotated by a slowly varying angle."""
    lo = torch.rand(B, 3, max(H // 8, 1), max(W // 8, 1), device=dev)
    f = torch.nn.functional.interpolate(lo, size=(H, W), mode="bilinear", align_corners=False)
    ang = (f[:, 0] - 0.5) * 3.14
    bw, bh = 2.0 + 4.0 * f[:, 1], 2.0 + 4.0 * f[:, 2]
    k = torch.arange(-1, 2, device=dev, dtype=torch.float32)
    ky, kx = torch.meshgrid(k, k, indexing="ij")           # (3,3) kernel grid
    ky, kx = ky.reshape(9, 1, 1, 1).transpose(0, 1), kx.reshape(9, 1, 1, 1).transpose(0, 1)  # (1,9,1,1)
    dx, dy = bw[:, None] / 3 * kx, bh[:, None] / 3 * ky
    c, s_ = torch.cos(ang)[:, None], torch.sin(ang)[:, None]
    xr, yr = c * dx - s_ * dy, s_ * dx + c * dy
    off = torch.stack([yr - ky, xr - kx], dim=2).reshape(B, 18, H, W)
    return off.contiguous()


def bench_old(B, C, O, H, W):
    x = torch.randn(B, C, H, W, device=dev)
    w_flat = (torch.randn(O, 9 * C, device=dev) / 48).bfloat16()
    off = anchor_offsets(B, H, W) if REAL else torch.randn(B, 18, H, W, device=dev)
    out = torch.empty((B, O, H * W), dtype=torch.bfloat16, device=dev)

    def f():
        col = dcn_v1.deformable_im2col(x, off, (3, 3), (1, 1), (1, 1), (1, 1), 1, col_dtype=torch.bfloat16)
        torch.bmm(w_flat.unsqueeze(0).expand(B, O, 9 * C), col.view(9 * C, B, H * W).permute(1, 0, 2), out=out)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f"old path (im2col bf16col + bmm) {H}x{W}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")


def bench(B, C, O, H, W, want_col):
    x_nhwc = torch.randn(B, H, W, C, device=dev).bfloat16()
    w_flat = (torch.randn(O, 9 * C, device=dev) / 48).bfloat16()
    off = anchor_offsets(B, H, W) if REAL else torch.randn(B, 18, H, W, device=dev)
    if os.environ.get("ACM_ZERO_OFF"):
        off = off * 0
    for _ in range(3):
        run(x_nhwc, off, w_flat, O, True, want_col)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        run(x_nhwc, off, w_flat, O, True, want_col)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    fl = 2.0 * B * H * W * O * 9 * C
    print(f"bench B{B} C{C} O{O} {H}x{W} col={want_col}: {us:.1f} us  {fl / us / 1e6:.1f} TFLOP/s")


if __name__ == "__main__":
    check32(2, 32, 32, 9, 13, 1.5)
    check32(1, 96, 96, 20, 31, 3.0)
    check32(2, 256, 256, 16, 16, 1.0)
    bench32(4, 256, 256, 128, 128, False)
    bench32(4, 256, 256, 128, 128, True)
    bench32(4, 256, 256, 64, 64, True)
    bench32(4, 256, 256, 8, 8, True)
    check(2, 64, 32, 9, 13, 1.5)
    check(1, 128, 96, 20, 31, 3.0)
    check(2, 256, 256, 16, 16, 1.0)
    for hw in (128, 64, 32, 16, 8):
        bench(4, 256, 256, hw, hw, False)
    bench(4, 256, 256, 128, 128, True)
    bench_old(4, 256, 256, 128, 128)
    bench_old(4, 256, 256, 64, 64)
    bench_old(4, 256, 256, 32, 32)
    bench_old(4, 256, 256, 16, 16)
    bench_old(4, 256, 256, 8, 8)
    REAL = False
    bench(4, 256, 256, 128, 128, False)
