import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rs_detection_amd import ops
from rs_detection_amd.utils import synthetic as syn
import bench
dev=torch.device('cuda')
anchors=torch.from_numpy(syn.s2anet_anchor_grid()).to(dev); A=anchors.shape[0]
tg=syn.synthetic_targets(4)
ks=[t["rboxes"].shape[0] for t in tg]
gt=torch.cat([torch.from_numpy(t["rboxes"]) for t in tg]).to(dev); lab=torch.cat([torch.from_numpy(t["labels"]) for t in tg]).to(dev).int()
ro=torch.tensor(np.concatenate([[0],np.cumsum(ks)]),dtype=torch.int32,device=dev); n1=sum(ks)
ov=ops.box_iou_rotated_grouped(gt,ro,max(ks),anchors)
t=bench.event_time(lambda: ops.assign_wrt_overlaps(ov,ro,max(ks),0.5,0.4,0.0,True,True,lab,0),50)
print(os.environ.get("RSDET_LIB_PATH","default").split("/")[-1], "assign %.1f us"%(t*1e6))
