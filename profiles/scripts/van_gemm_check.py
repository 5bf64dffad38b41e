"""rsdet_van_gemm_f32 on the VAN-B3 shapes of the 2 x 1024^2 Oriented R-CNN step: values against torch (fp64 reference of
the fp32 operands) for every epilogue, time against torch.matmul / F.conv2d (rocBLAS / MIOpen) of the same product."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from rs_detection_amd import _lib

lib = _lib.load()
dev = torch.device("cuda")
torch.manual_seed(0)


def gemm(w, x, epi=0, v=(None,) * 4, s=(None, None), two=False):
    n, K, P = x.shape
    M = w.shape[0]
    o0 = torch.empty((n, M, P), device=dev)
    o1 = torch.empty((n, M, P), device=dev) if two else None
    rc = lib.rsdet_van_gemm_f32(_lib.ptr(w), _lib.ptr(x), M, K, P, n, epi, *[_lib.ptr(t) for t in v], *[_lib.ptr(t) for t in s],
                                _lib.ptr(o0), _lib.ptr(o1), _lib.stream_ptr())
    _lib.check(rc, "rsdet_van_gemm_f32")
    return o0, o1


def t_us(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def gelu(x):
    return F.gelu(x)


def gelu_grad(x):
    x = x.double()
    return 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5


shapes = [(64, 64, 65536), (512, 64, 65536), (64, 512, 65536), (128, 128, 16384), (1024, 128, 16384), (128, 1024, 16384),
          (320, 320, 4096), (1280, 320, 4096), (320, 1280, 4096), (512, 512, 1024), (2048, 512, 1024), (512, 2048, 1024)]
n = 2
tot_own = tot_lib = 0.0
for M, K, P in shapes:
    assert lib.rsdet_van_gemm_f32_supported(M, K, P, n), (M, K, P)
    w = torch.randn(M, K, device=dev) / K ** 0.5
    x = torch.randn(n, K, P, device=dev)
    v = [torch.randn(M, device=dev) for _ in range(4)]
    s0, s1 = torch.randn(n, M, P, device=dev), torch.randn(n, M, P, device=dev)
    ref = torch.matmul(w.double(), x.double())
    scale = float(ref.abs().max())
    col = lambda t: t.double()[None, :, None]
    errs = []
    o0, _ = gemm(w, x)
    errs.append(float((o0 - ref).abs().max()) / scale)
    o0, _ = gemm(w, x, 1, (v[0], None, None, None))
    errs.append(float((o0 - (ref + col(v[0]))).abs().max()) / scale)
    o0, o1 = gemm(w, x, 2, (v[0], None, None, None), two=True)
    errs.append(max(float((o0 - (ref + col(v[0]))).abs().max()), float((o1 - gelu(ref + col(v[0]))).abs().max())) / scale)
    o0, o1 = gemm(w, x, 3, (v[0], None, None, None), (s0, None), two=True)
    errs.append(max(float((o0 - (ref + col(v[0]))).abs().max()), float((o1 - (ref + col(v[0])) * s0.double()).abs().max())) / scale)
    o0, _ = gemm(w, x, 4, v, (s0, s1))
    want = s0.double() * col(v[0]) + ref * col(v[1]) + col(v[2]) + s1.double() * col(v[3])
    errs.append(float((o0 - want).abs().max()) / float(want.abs().max()))
    o0, _ = gemm(w, x, 4, (None, v[1], v[2], None), (s0, None))
    want = s0.double() + ref * col(v[1]) + col(v[2])
    errs.append(float((o0 - want).abs().max()) / float(want.abs().max()))
    o0, o1 = gemm(w, x, 5, s=(s0, s1), two=True)
    errs.append(max(float((o0 - ref * s0.double()).abs().max()), float((o1 - ref * s1.double()).abs().max())) / scale / 4)
    o0, _ = gemm(w, x, 6, s=(s0, None))
    errs.append(float((o0 - ref * s0.double()).abs().max()) / scale / 4)
    o_f32, _ = gemm(w, x)
    denom = float((w.double().abs() @ x.double().abs()).max())          # sum |a_k b_k|: what the error bound is relative to
    e_f32 = float((o_f32 - ref).abs().max()) / denom
    us = t_us(lambda: gemm(w, x))
    us2 = t_us(lambda: gemm(w, x, 2, (v[0], None, None, None), two=True))
    us4 = t_us(lambda: gemm(w, x, 4, v, (s0, s1)))
    w4 = w.view(M, K, 1, 1)
    x4 = x.view(n, K, int(P ** 0.5), -1)
    us_lib = t_us(lambda: F.conv2d(x4, w4))
    fl = 2.0 * M * K * P * n
    tot_own += us
    tot_lib += us_lib
    print("M %4d K %4d P %5d: max err / scale %.1e | plain %6.1f us (%5.1f TF/s, %.2f of 157.3)  gelu2 %6.1f  affine %6.1f | "
          "MIOpen conv2d %6.1f us | max |err| / sum|ab| %.1e" % (M, K, P, max(errs), us, fl / us / 1e6, fl / us / 1e6 / 157.3, us2, us4,
                                                                    us_lib, e_f32))
    assert max(errs) < 1e-5, errs
print("sum over the 12 shapes: own %.1f us, library %.1f us" % (tot_own, tot_lib))
