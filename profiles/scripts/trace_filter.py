import sys, os, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rs_detection_amd import _lib
from rs_detection_amd.utils import synthetic as syn
dev=torch.device('cuda'); lib=_lib.load()
rng=np.random.default_rng(0)
a=torch.from_numpy(syn.s2anet_anchor_grid()).to(dev)
K=int(sys.argv[1]) if len(sys.argv)>1 else 400
g=torch.from_numpy(syn.dota_gt_boxes(rng,K)).to(dev)
n1,n2=K,a.shape[0]
out=torch.empty((n1,n2),device=dev)
wsb=lib.rsdet_box_iou_rotated_ws_size(n1,n2,n2); ws=torch.empty(wsb,dtype=torch.uint8,device=dev)
TI=int(os.environ.get("TI","32")); nb=((n2+63)//64)*((n1+TI-1)//TI)
tr=torch.zeros(nb*8,dtype=torch.int64,device=dev)
def call(): lib.rsdet_box_iou_rotated_f32(_lib.ptr(g),n1,5,_lib.ptr(a),n2,5,0,_lib.ptr(out),_lib.ptr(ws),wsb,_lib.stream_ptr())
for _ in range(5): call()
torch.cuda.synchronize()
lib.rsdet_debug_set_trace.argtypes=[ctypes.c_void_p]; lib.rsdet_debug_set_trace(ctypes.c_void_p(tr.data_ptr()))
torch.cuda.synchronize()
call(); torch.cuda.synchronize()
raw=tr.cpu().numpy().reshape(nb,8)
t=raw[:,:6].astype(np.float64)*0.01  # us
t0=t[:,0].min()
print("waves",nb,"span %.2f us"%(t[:,:5].max()-t0))
print("start pct 10/50/90/100 = %s"%np.percentile(t[:,0]-t0,[10,50,90,100]).round(2))
names=["loads+cull","circle+SAT rows","scan+atomic+queue","bitmap+zero fill issue"]
for k in range(4):
    ok=(t[:,k+1]>0)
    d=(t[ok,k+1]-t[ok,k])
    print("%-24s n=%d mean %.2f  p50 %.2f p90 %.2f max %.2f"%(names[k],ok.sum(),d.mean(),*np.percentile(d,[50,90,100])))
end=t.max(axis=1)-t0
print("end pct 10/50/90/100 = %s"%np.percentile(end,[10,50,90,100]).round(2))
live=raw[:,6]; tot=raw[:,7]
print("live strips per wave: mean %.2f ; hist %s"%(live.mean(), np.bincount(live.astype(int),minlength=TI+1)))
print("survivors per wave: mean %.1f max %d; total %d"%(tot.mean(), tot.max(), tot.sum()))
