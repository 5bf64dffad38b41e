import sys, os, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rs_detection_amd import _lib
from rs_detection_amd.utils import synthetic as syn
dev=torch.device('cuda'); lib=_lib.load()
rng=np.random.default_rng(0)
a=torch.from_numpy(syn.s2anet_anchor_grid()).to(dev)
K=int(sys.argv[1]) if len(sys.argv)>1 else 400
TI=int(os.environ.get("TI","16"))
g=torch.from_numpy(syn.dota_gt_boxes(rng,K)).to(dev)
n1,n2=K,a.shape[0]
out=torch.empty((n1,n2),device=dev)
wsb=lib.rsdet_box_iou_rotated_ws_size(n1,n2,n2); ws=torch.empty(wsb,dtype=torch.uint8,device=dev)
nb=((n2+255)//256)*((n1+TI-1)//TI)
tr=torch.zeros(nb*8,dtype=torch.int64,device=dev)
def call(): lib.rsdet_box_iou_rotated_f32(_lib.ptr(g),n1,5,_lib.ptr(a),n2,5,0,_lib.ptr(out),_lib.ptr(ws),wsb,_lib.stream_ptr())
for _ in range(5): call()
torch.cuda.synchronize()
lib.rsdet_debug_set_trace.argtypes=[ctypes.c_void_p]; lib.rsdet_debug_set_trace(ctypes.c_void_p(tr.data_ptr()))
torch.cuda.synchronize()
call(); torch.cuda.synchronize()
t=tr.cpu().numpy().reshape(nb,8).astype(np.float64)*0.01
t0=t[:,0].min()
print("blocks",nb,"kernel span %.2f us"%(t[:,:7].max()-t0))
print("block start pct 10/50/90/100 = %s"%np.percentile(t[:,0]-t0,[10,50,90,100]).round(2))
names=["load+LDS stage","fill issue+barrier","cull+passA","passB (SAT)","atomic+barrier","late fill+queue write"]
for k in range(6):
    ok=(t[:,k+1]>0)&(t[:,k]>0); d=(t[ok,k+1]-t[ok,k])
    if ok.sum(): print("%-24s n=%d mean %.2f p50 %.2f p90 %.2f max %.2f"%(names[k],ok.sum(),d.mean(),*np.percentile(d,[50,90,100])))
end=t[:,:7].max(axis=1)-t0
print("block end pct 10/50/90/100 = %s"%np.percentile(end,[10,50,90,100]).round(2))
