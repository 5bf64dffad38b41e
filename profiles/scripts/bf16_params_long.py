import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from rs_detection_amd.runner.runner import Runner
dev = torch.device("cuda", 0)
mode = sys.argv[1] == "1"
torch.manual_seed(0)
r = Runner(bench.s2anet_cfg(), device=dev, memory_format=torch.channels_last, amp_dtype=torch.bfloat16, bf16_params=mode)
batches = bench.make_batches(4, 4, 0, 15, dev, torch.channels_last, False)
out = []
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    t, p = r.train_step(*batches[i % 4])
    out.append(float(t))
    if not np.isfinite(out[-1]):
        print("NON-FINITE at step", i, {k: float(v) for k, v in p.items()}); break
print("bf16_params", mode, "losses", np.round(out[:3], 3), "...", np.round(out[-6:], 3))
