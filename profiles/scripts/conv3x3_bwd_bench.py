import sys, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from rs_detection_amd.utils.miopen_db import use_packaged_miopen_db
use_packaged_miopen_db()
dev = torch.device("cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
shapes = [(4, 256, 128, 128), (4, 256, 128, 196), (4, 256, 64, 100), (4, 256, 64, 64), (4, 64, 256, 256), (4, 128, 128, 128), (4, 256, 64, 64), (4, 512, 32, 32)]
for dt in (torch.bfloat16, torch.float32):
    for B, C, H, W in shapes:
        x = torch.randn(B, C, H, W, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(C, C, 3, 3, device=dev) * 0.02).to(dt).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(B, C, H, W, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        wt = w.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last)
        f = t(lambda: F.conv2d(x, w, None, 1, 1))
        b = t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (True, False, False)))
        bf = t(lambda: F.conv2d(gy, wt, None, 1, 1))
        fl = t(lambda: w.flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last))
        g1 = torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (True, False, False))[0].float()
        g2 = F.conv2d(gy, wt, None, 1, 1).float()
        err = float((g1 - g2).abs().max() / g1.abs().max())
        print("%s B=%d C=%d %dx%d: fwd %.1f | bwd-data %.1f | bwd-data as fwd %.1f (+ flip %.1f) us | rel diff %.1e" % (str(dt)[6:], B, C, H, W, f, b, bf, fl, err))
