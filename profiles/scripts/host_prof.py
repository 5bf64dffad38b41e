import sys, os, cProfile, pstats, io; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
from rs_detection_amd.utils.miopen_db import use_packaged_miopen_db; use_packaged_miopen_db()
import bench
from rs_detection_amd.runner.runner import Runner
from rs_detection_amd.utils import synthetic as syn
dev = torch.device("cuda", 0)
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
runner = Runner(bench.s2anet_cfg(), device=dev, amp_dtype=torch.bfloat16 if dt == "bf16" else None,
                memory_format=torch.channels_last if dt == "bf16" else None, bf16_params=(dt == "bf16"))
images = torch.randn(4, 3, 1024, 1024, device=dev)
if dt == "bf16":
    images = images.contiguous(memory_format=torch.channels_last)
targets = []
for t in syn.synthetic_targets(4, rank=0, it=0, num_classes=15):
    t = dict(t); t["rboxes"] = torch.from_numpy(t["rboxes"]).to(dev); t["labels"] = torch.from_numpy(t["labels"]).to(dev); targets.append(t)
for _ in range(5): runner.train_step(images, targets)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): runner.train_step(images, targets)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(40); print(s.getvalue()[:9000])
