"""Host-side profile of the Oriented R-CNN VAN-B3 step: where the Python thread spends its time (cProfile) and how long
it is BLOCKED on the device (the nonzero() calls of the samplers synchronise).  python profiles/scripts/host_prof_orcnn.py"""
import sys, os, cProfile, pstats, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
from rs_detection_amd.utils.miopen_db import use_packaged_miopen_db; use_packaged_miopen_db()
import bench
from rs_detection_amd.config import Config
from rs_detection_amd.runner.runner import Runner
dev = torch.device("cuda", 0)
cfg = Config(os.path.join(ROOT, "configs", "orcnn", "orcnn_van3_7_anchor.py"))
runner = Runner(cfg, device=dev)
batches = bench.make_batches(4, 2, 0, 10, dev, None, True)
for b in batches: runner.train_step(*b)
for i in range(3): runner.train_step(*batches[i % 4])
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
for i in range(8): runner.train_step(*batches[i % 4])
pr.disable(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("enqueue %.2f ms/step, total %.2f ms/step" % ((t1 - t0) / 8 * 1e3, (t2 - t0) / 8 * 1e3))
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(45); print(s.getvalue()[:12000])
