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
B = 4
shapes = [(64, 64, 256), (256, 64, 256), (64, 256, 256), (256, 128, 256), (512, 128, 128), (128, 512, 128), (512, 256, 128),
          (1024, 256, 64), (256, 1024, 64), (1024, 512, 64), (2048, 512, 32), (512, 2048, 32), (256, 256, 128), (512, 256, 64), (1024, 256, 32), (2048, 256, 16)]
for dt in (torch.bfloat16, torch.float32):
    tot = [0, 0]
    for C, O, H in shapes:
        x = torch.randn(B, C, H, H, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(O, C, 1, 1, device=dev) * 0.05).to(dt).contiguous(memory_format=torch.channels_last)
        gy = torch.randn(B, O, H, H, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
        x2, gy2, w2 = x.permute(0, 2, 3, 1).reshape(-1, C), gy.permute(0, 2, 3, 1).reshape(-1, O), w.view(O, C)
        f_c = t(lambda: F.conv2d(x, w))
        f_m = t(lambda: torch.mm(x2, w2.t()))
        d_c = t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (True, False, False)))
        d_m = t(lambda: torch.mm(gy2, w2))
        w_c = t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (False, True, False)))
        w_m = t(lambda: torch.mm(gy2.t(), x2))
        y1, y2 = F.conv2d(x, w).permute(0, 2, 3, 1).reshape(-1, O).float(), torch.mm(x2, w2.t()).float()
        err = float((y1 - y2).abs().max() / (y1.abs().max() + 1e-9))
        tot[0] += f_c + d_c + w_c; tot[1] += f_m + d_m + w_m
        print("%s C=%4d O=%4d H=%3d  fwd conv %6.1f mm %6.1f | bwd-data conv %6.1f mm %6.1f | wrw conv %6.1f mm %6.1f us | rel diff %.1e" % (
            str(dt)[6:], C, O, H, f_c, f_m, d_c, d_m, w_c, w_m, err))
    print("sum", tot)
