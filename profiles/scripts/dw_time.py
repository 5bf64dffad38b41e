import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from rs_detection_amd.ops.dwconv import dwconv2d
dev = torch.device('cuda')
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for (N, C, H, K, D) in [(2, 512, 256, 3, 1), (2, 64, 256, 5, 1), (2, 64, 256, 7, 3), (2, 1024, 128, 3, 1), (2, 128, 128, 7, 3),
                        (2, 1280, 64, 3, 1), (2, 320, 64, 7, 3), (2, 2048, 32, 3, 1), (2, 512, 32, 7, 3)]:
    x = torch.randn(N, C, H, H, device=dev, requires_grad=True)
    w = torch.randn(C, 1, K, K, device=dev, requires_grad=True)
    b = torch.randn(C, device=dev, requires_grad=True)
    p = D * (K - 1) // 2
    mb = x.numel() * 4 / 1e6
    res = []
    for name, f in (("hip", lambda: dwconv2d(x, w, b, D)), ("torch", lambda: F.conv2d(x, w, b, 1, p, D, C))):
        y = f(); go = torch.randn_like(y)
        tf = t(f)
        tb = t(lambda: torch.autograd.grad(f(), (x, w, b), go)) - tf
        res.append((name, tf, tb))
    print("N%d C%4d H%3d K%d D%d  tensor %6.1f MB | " % (N, C, H, K, D, mb) +
          " | ".join("%s fwd %7.1f us (%4.2f TB/s) bwd %7.1f us" % (n, a, 2 * mb / a, bb) for n, a, bb in res))
