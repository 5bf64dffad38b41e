import sys, os, torch
sys.path.insert(0, "/root/repo")
from rs_detection_amd import _lib
lib = _lib.load()
dev = torch.device("cuda")
def gt(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
out = []
for C, H, res in ((256, 256, True), (64, 256, False), (512, 128, True), (128, 128, False), (1024, 64, True), (256, 64, False), (2048, 32, True)):
    N = 4
    mk = lambda: torch.randn(N, H, H, C, device=dev).bfloat16()
    gy, y, x, gx = mk(), mk(), mk(), mk()
    gres = mk() if res else None
    mean, var, w = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.ones(C, device=dev)
    gw, gb = torch.empty(C, device=dev), torch.empty(C, device=dev)
    wsb = lib.rsdet_bn_act_backward_nhwc_ws_size(N, C, H * H)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    def call():
        rc = lib.rsdet_bn_act_backward_nhwc_bf16(_lib.ptr(gy), _lib.ptr(y), _lib.ptr(x), _lib.ptr(mean), _lib.ptr(var), _lib.ptr(w), 1e-5,
                                                 N, C, H * H, 1, _lib.ptr(gx), _lib.ptr(gres), _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(ws), wsb,
                                                 _lib.stream_ptr())
        assert rc == 0
    t = gt(call)
    nbytes = gy.numel() * 2 * (5 if res else 4)
    out.append("C=%d H=%d res=%d: %.1f us (%.2f TB/s)" % (C, H, res, t, nbytes / t / 1e6))
print(os.path.basename(os.environ.get("RSDET_LIB_PATH", "default")), "cap", os.environ.get("RSDET_BN_SCAP", "-"), " | ".join(out))
