import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from rs_detection_amd.utils.miopen_db import use_packaged_miopen_db
use_packaged_miopen_db()
import rs_detection_amd.models
from rs_detection_amd.config import Config
from rs_detection_amd.runner.runner import Runner
from rs_detection_amd.utils.synthetic import synthetic_targets
cfg = Config("/root/repo/configs/s2anet/s2anet_r50_fpn_1x_dota.py")
dev = torch.device("cuda")
for amp, mf in ((None, None), (None, torch.channels_last), (torch.bfloat16, torch.channels_last)):
    r = Runner(cfg, device=dev, memory_format=mf, amp_dtype=amp)
    for B in (1, 4):
        im = torch.randn(B, 3, 1024, 1024, device=dev)
        tg = synthetic_targets(B)
        for packed in ("1", "0"):
            os.environ["RSDET_S2A_PACKED"] = packed
            for _ in range(3): r.predict(im, tg)
            torch.cuda.synchronize(); t = time.time()
            for _ in range(10): r.predict(im, tg)
            torch.cuda.synchronize()
            print("amp", amp, "mf", mf, "B", B, "packed", packed, "predict %.2f ms" % ((time.time() - t) * 100))
