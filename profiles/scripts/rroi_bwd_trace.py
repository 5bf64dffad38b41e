import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
from rs_detection_amd import ops
from rs_detection_amd.utils import synthetic as syn
dev = torch.device("cuda"); rng = np.random.default_rng(3)
N, C, H, R = 2, 256, 256, 512
feat = torch.randn(N, C, H, H, device=dev, requires_grad=True)
b = syn.dota_gt_boxes(rng, R).astype(np.float32)
rois = torch.from_numpy(np.concatenate([rng.integers(0, N, (R, 1)).astype(np.float32), b], 1)).to(dev)
y = ops.roi_align_rotated_v1(feat, rois, (7, 7), 0.25, 2); go = torch.randn_like(y)
for _ in range(6):
    torch.autograd.grad(y, feat, go, retain_graph=True)
torch.cuda.synchronize()
