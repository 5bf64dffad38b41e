import sys, os, ctypes; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rs_detection_amd import ops, _lib
from rs_detection_amd.utils import synthetic as syn
dev=torch.device('cuda'); lib=_lib.load()
M=int(sys.argv[1]) if len(sys.argv)>1 else 5344
d, s, l = syn.nms_cluster_boxes(M)
d6 = torch.from_numpy(np.concatenate([d, l[:, None].astype(np.float32)], 1)).to(dev)
order = torch.from_numpy(np.argsort(-s, kind="stable").astype(np.int32)).to(dev)
f = lambda: ops.nms_rotated_keep_mask(d6, order, 0.1, 6)
for _ in range(3): f()
torch.cuda.synchronize()
cb=(M+63)//64
tr=torch.zeros(cb*8,dtype=torch.int64,device=dev)
lib.rsdet_debug_set_sweep_trace.argtypes=[ctypes.c_void_p]; lib.rsdet_debug_set_sweep_trace(ctypes.c_void_p(tr.data_ptr()))
torch.cuda.synchronize(); f(); torch.cuda.synchronize()
raw=tr.cpu().numpy().reshape(cb,8); t=raw[:,:4].astype(np.float64)*0.01
names=["prefetch+decide","store+barrier1","apply+barrier2"]
print("steps",cb,"total %.1f us, per step %.2f"%(t[-1,3]-t[0,0],(t[-1,3]-t[0,0])/cb))
for k in range(3):
    dlt=t[:,k+1]-t[:,k]; ok=(t[:,k+1]>0)&(t[:,k]>0)
    print("%-22s mean %.3f p50 %.3f max %.3f"%(names[k],dlt[ok].mean(),np.median(dlt[ok]),dlt[ok].max()))
gap=t[1:,0]-t[:-1,3]; print("between steps mean %.3f"%gap.mean())

