import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from torch.utils.flop_counter import FlopCounterMode
import bench
from rs_detection_amd.runner.runner import Runner
from rs_detection_amd.utils import synthetic as syn
dev=torch.device("cuda")
torch.manual_seed(0)
r=Runner(bench.s2anet_cfg(), device=dev, distributed=False)
images=torch.randn(4,3,1024,1024,device=dev)
targets=[]
for t in syn.synthetic_targets(4, rank=0, it=0):
    t=dict(t); t["rboxes"]=torch.from_numpy(t["rboxes"]).to(dev); t["labels"]=torch.from_numpy(t["labels"]).to(dev); targets.append(t)
r.train_step(images,targets)
with FlopCounterMode(display=False) as fc:
    r.train_step(images,targets)
tot=fc.get_total_flops()
print("total flops per step (4 tiles): %.3f TFLOP"%(tot/1e12))
for k,v in sorted(fc.get_flop_counts()["Global"].items(), key=lambda kv:-kv[1])[:8]: print("  ",k, "%.3f"%(v/1e12))
