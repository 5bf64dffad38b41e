import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rs_detection_amd.ops.box_iou_rotated import _iou
G=os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),"tests","golden")
dev=torch.device("cuda")
d=np.load(os.path.join(G,"iou_v0.npz"))
for rep in range(3):
    for name,(a,b,want) in {"rand":(d["boxes1"],d["boxes2"],d["ious"]),"anchor":(d["gts"],d["anchors"],d["ious_anchor"])}.items():
        ta=torch.from_numpy(np.ascontiguousarray(a)).to(dev); tb=torch.from_numpy(np.ascontiguousarray(b)).to(dev)
        got=_iou(ta,tb,0).cpu().numpy()
        w=np.argwhere(~(np.abs(got-want)<=1e-4))
        print("rep",rep,name,got.shape,"bad",len(w))
        for i,j in w[:8]:
            print("   pair",i,j,"got",got[i,j],"want",want[i,j],"box1",a[i],"box2",b[j])
        if len(w):
            got2=_iou(ta,tb,0).cpu().numpy()
            print("   recompute bad:",int((~(np.abs(got2-want)<=1e-4)).sum()))
