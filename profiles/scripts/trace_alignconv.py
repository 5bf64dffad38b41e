"""Per-step timestamps of workgroup 0 of alignconv_fwd_mfma_kernel (debug build -DACM_TRACE, scratch/lib_acmtrace.so):
RSDET_LIB_PATH=scratch/lib_acmtrace.so python profiles/scripts/trace_alignconv.py"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rs_detection_amd import _lib
dev = torch.device("cuda")
lib = _lib.load()
B, C, O, H, W = 4, 256, 256, int(os.environ.get("HW", 128)), int(os.environ.get("HW", 128))
x = torch.randn(B, H, W, C, device=dev).bfloat16()
w = (torch.randn(O, 9 * C, device=dev) / 48).bfloat16()
off = torch.zeros(B, 18, H, W, device=dev)
out = torch.empty((B, H, W, O), dtype=torch.bfloat16, device=dev)
g = _lib.DcnGeom(C, H, W, 3, 3, 1, 1, 1, 1, 1, 1, B, 1)
tr = torch.zeros(2 * 80 * 4, dtype=torch.int64, device=dev)
f = lib.rsdet_debug_set_acm_trace
f.argtypes = [ctypes.c_void_p]
def call():
    assert lib.rsdet_alignconv_fwd_mfma_bf16(_lib.ptr(x), _lib.ptr(off), _lib.ptr(w), g, O, 1, _lib.ptr(out), None, _lib.stream_ptr()) == 0
for _ in range(3):
    call()
torch.cuda.synchronize()
f(ctypes.c_void_p(tr.data_ptr()))
call()
torch.cuda.synchronize()
f(None)
t = tr.cpu().numpy().reshape(2, 80, 4).astype(np.float64)
t0 = t[0, 0, 0]
print("step: consumer [start, mfma done, B landed, barrier passed]  producer [start, done]   (cycles since step 0)")
for s in range(36):
    print(s, (t[0, s] - t0).astype(int), (t[1, s, :2] - t0).astype(int), " step time", int(t[0, s, 3] - t[0, s, 0]))
