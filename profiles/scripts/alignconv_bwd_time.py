import sys, torch
sys.path.insert(0, "/root/repo")
from rs_detection_amd.ops.dcn_v1 import deformable_col2im_gather_nhwc
dev = torch.device("cuda")
def gt(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
B, C, O = 4, 256, 256
for H in (128, 64, 32):
    P = B * H * H
    torch.manual_seed(0)
    go = torch.randn(P, O, device=dev).bfloat16()
    w = (torch.randn(O, 9 * C, device=dev) / 48).bfloat16()
    off = (torch.randn(B, 18, H, H, device=dev) * 1.5)
    gcol = torch.mm(go, w)
    t_mm = gt(lambda: torch.mm(go, w))
    t_g = gt(lambda: deformable_col2im_gather_nhwc(gcol, off, (B, H, H, C), (3, 3), (1, 1), (1, 1), (1, 1)))
    print("level H=%d: gcolT = go @ W (%d x 256 x 2304, bf16) %.1f us | col2im gather chain %.1f us | backward-data total %.1f us" % (H, P, t_mm, t_g, t_mm + t_g))
