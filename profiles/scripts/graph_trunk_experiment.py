"""Experiment: backbone + neck of S2ANet as torch.cuda.make_graphed_callables (forward + backward hipGraphs)."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch, numpy as np
from rs_detection_amd.utils.miopen_db import use_packaged_miopen_db; use_packaged_miopen_db()
import bench
from rs_detection_amd.runner.runner import Runner
from rs_detection_amd.utils import synthetic as syn
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
graphed = (sys.argv[2] == "1") if len(sys.argv) > 2 else True
dev = torch.device("cuda", 0)
cfg = bench.s2anet_cfg()
runner = Runner(cfg, device=dev, amp_dtype=torch.bfloat16 if dt == "bf16" else None)
images = torch.randn(4, 3, 1024, 1024, device=dev)
def targets(it):
    out = []
    for t in syn.synthetic_targets(4, rank=0, it=0, num_classes=15):
        t = dict(t)
        t["rboxes"] = torch.from_numpy(t["rboxes"]).to(dev)
        t["labels"] = torch.from_numpy(t["labels"]).to(dev)
        out.append(t)
    return out
m = runner.model
if graphed:
    class Trunk(torch.nn.Module):
        def __init__(s, b, n): super().__init__(); s.b, s.n = b, n
        def forward(s, x): return tuple(s.n(s.b(x)))
    trunk = Trunk(m.backbone, m.neck)
    m.train()
    sample = (images.clone().requires_grad_(False),)
    if dt == "bf16":
        with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
            g = torch.cuda.make_graphed_callables(trunk, sample)
    else:
        g = torch.cuda.make_graphed_callables(trunk, sample)
    def fwd(images, tg):
        return m.bbox_head(list(g(images)), tg)
    m.forward = fwd
for it in range(5): runner.train_step(images, targets(it))
torch.cuda.synchronize(); t0 = time.time()
N = 40
losses = []
for it in range(N):
    l, _ = runner.train_step(images, targets(5 + it))
    if it % 5 == 4: losses.append(float(l))
torch.cuda.synchronize(); dtm = (time.time() - t0) / N
print("dtype %s graphed %s: %.2f ms/step  losses %s" % (dt, graphed, dtm * 1e3, ["%.3f" % v for v in losses]))
