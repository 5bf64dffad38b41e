import sys, torch
sys.path.insert(0, "/root/repo")
from rs_detection_amd.ops.bn_act import bn_relu_maxpool, bn_act
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
bn = torch.nn.BatchNorm2d(64).to(dev).eval()
pool = torch.nn.MaxPool2d(3, 2, 1)
for dt in (torch.bfloat16, torch.float32):
    x = torch.randn(4, 64, 512, 512, device=dev).to(dt).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        a = gt(lambda: bn_relu_maxpool(x, bn, pool))
        b = gt(lambda: bn_act(x, bn))
        y = bn_act(x, bn)
        c = gt(lambda: pool(y))
    print(dt, "fused %.1f us | bn_act %.1f + maxpool %.1f us" % (a, b, c))
