import sys, os, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rs_detection_amd import _lib
L=ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)),'libfloor.so'))
vp=ctypes.c_void_p; ll=ctypes.c_longlong
dev=torch.device('cuda')
def gtime(fn, n=20, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/(n*reps)*1e3
print("null kernel in graph: %.2f us"%gtime(lambda: L.run_null(_lib.stream_ptr())))
for mb in (35, 48.6, 97, 400):
    n=int(mb*1e6)//16*16; buf=torch.empty(n,dtype=torch.uint8,device=dev); p=vp(buf.data_ptr())
    r=[]
    r.append(("memset", gtime(lambda: L.run_memset(p, ll(n), _lib.stream_ptr()))))
    r.append(("torch.zero_", gtime(lambda: buf.zero_())))
    for blocks in (1024, 2048, 4096, 8192, 16384):
        r.append(("fill4 g%d"%blocks, gtime(lambda: L.run_fill4(p, ll(n//16), blocks, _lib.stream_ptr()))))
    for pb in (1024, 4096, 16384):
        r.append(("fill4_tile pb%d"%pb, gtime(lambda: L.run_fill4_tile(p, ll(n//16), pb, _lib.stream_ptr()))))
    r.append(("fill1 g8192", gtime(lambda: L.run_fill1(p, ll(n//4), 8192, _lib.stream_ptr()))))
    print("%.1f MB: "%mb + "  ".join("%s %.1fus(%.2fTB/s)"%(k,t,n/t/1e6) for k,t in r))
