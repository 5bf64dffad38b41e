import sys, os, time, torch, numpy as np
sys.path.insert(0, ".")
import bench
from rs_detection_amd.runner.runner import Runner
from rs_detection_amd.utils import synthetic as syn
dev = torch.device("cuda", 0)
torch.manual_seed(0)
dt = sys.argv[1]
bfp = len(sys.argv) > 2 and sys.argv[2] == "bf16params"
r = Runner(bench.s2anet_cfg(), device=dev, memory_format=torch.channels_last if dt == "bf16" else None, amp_dtype=torch.bfloat16 if dt == "bf16" else None, bf16_params=bfp)
g = torch.Generator(device="cpu").manual_seed(0)
images = torch.randn(4, 3, 1024, 1024, generator=g).to(dev)
if dt == "bf16": images = images.contiguous(memory_format=torch.channels_last)
targets = []
for t in syn.synthetic_targets(4, rank=0, it=0, num_classes=15, img=1024):
    t = dict(t); t["rboxes"] = torch.from_numpy(t["rboxes"]).to(dev); t["labels"] = torch.from_numpy(t["labels"]).to(dev); targets.append(t)
for i in range(5): r.train_step(images, targets)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(30): r.train_step(images, targets)
t1 = time.perf_counter()
torch.cuda.synchronize(); t2 = time.perf_counter()
print(dt, "bf16_params" if bfp else "fp32 params", "host enqueue %.2f ms/step, total %.2f ms/step, GPU tail after last enqueue %.2f ms" % ((t1 - t0) / 30 * 1e3, (t2 - t0) / 30 * 1e3, (t2 - t1) * 1e3))
