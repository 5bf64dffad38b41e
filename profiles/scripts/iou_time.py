import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rs_detection_amd import ops, _lib
from rs_detection_amd.utils import synthetic as syn
dev=torch.device('cuda'); lib=_lib.load()
rng=np.random.default_rng(0)
a=torch.from_numpy(syn.s2anet_anchor_grid()).to(dev)
def run(K, reps=200):
    g=torch.from_numpy(syn.dota_gt_boxes(rng,K)).to(dev)
    n1,n2=K,a.shape[0]
    out=torch.empty((n1,n2),device=dev)
    wsb=lib.rsdet_box_iou_rotated_ws_size(n1,n2,n2); ws=torch.empty(wsb,dtype=torch.uint8,device=dev)
    st=_lib.stream_ptr()
    def call(): lib.rsdet_box_iou_rotated_f32(_lib.ptr(g),n1,5,_lib.ptr(a),n2,5,0,_lib.ptr(out),_lib.ptr(ws),wsb,_lib.stream_ptr())
    for _ in range(10): call()
    torch.cuda.synchronize()
    gr=torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20): call()
    gr.replay(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps//20): gr.replay()
    e.record(); torch.cuda.synchronize()
    t=s.elapsed_time(e)/reps*1e-3
    print(f"K={K}: {t*1e6:.1f} us per call (3 kernels, graph)  {K*n2/t/1e6:.0f} Mpairs/s  {(4*K*n2)/t/1e9:.0f} GB/s")
for K in ([int(x) for x in sys.argv[1:]] or (16,100,400,1600)): run(K)
