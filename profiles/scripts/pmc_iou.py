import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rs_detection_amd import ops
from rs_detection_amd.utils import synthetic as syn
dev=torch.device('cuda')
rng=np.random.default_rng(0)
a=torch.from_numpy(syn.s2anet_anchor_grid()).to(dev)
ks=[16,100,400,40]
g=torch.from_numpy(np.concatenate([syn.dota_gt_boxes(rng,k) for k in ks])).to(dev)
ro=torch.tensor(np.concatenate([[0],np.cumsum(ks)]),dtype=torch.int32,device=dev)
out=torch.empty((sum(ks),a.shape[0]),device=dev)
lab=torch.ones(sum(ks),dtype=torch.int32,device=dev)
for _ in range(5):
    ops.box_iou_rotated_grouped(g,ro,max(ks),a,out=out)
    ops.assign_wrt_overlaps(out,ro,max(ks),0.5,0.4,0.0,True,True,lab,0)
B,C,H=4,256,128
x=torch.randn(B,C,H,H,device=dev); off=torch.randn(B,18,H,H,device=dev)
for _ in range(3):
    col=ops.deformable_im2col(x,off,(3,3),(1,1),(1,1),(1,1))
colT=torch.randn(B*H*H,9*C,device=dev)
for _ in range(3):
    ops.deformable_col2im_nhwc(colT,off,(B,H,H,C),(3,3),(1,1),(1,1),(1,1))
torch.cuda.synchronize(); print("done")
