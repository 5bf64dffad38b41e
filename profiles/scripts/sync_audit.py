"""Which statements of a train step synchronise the host with the device?  torch.cuda.set_sync_debug_mode("warn") flags
every synchronising call (.item(), nonzero, a host->device copy of a scalar / pageable buffer ...); this script runs
one warmed-up step under it and prints the statements of this repo the warnings come from.
  python profiles/scripts/sync_audit.py s2anet [f32|bf16]      (round 4: none)
  python profiles/scripts/sync_audit.py orcnn                   (round 4: 8 per step -- 4 sampler counts, 2 proposal
                                                                 validity checks, 2 NMS keep lists; 30 at its start)"""
import collections
import os
import sys
import traceback
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch  # noqa: E402
from rs_detection_amd.utils.miopen_db import use_packaged_miopen_db  # noqa: E402
use_packaged_miopen_db()
import bench  # noqa: E402
from rs_detection_amd.config import Config  # noqa: E402
from rs_detection_amd.runner.runner import Runner  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "s2anet"
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda", 0)
if model == "orcnn":
    runner = Runner(Config(os.path.join(ROOT, "configs", "orcnn", "orcnn_van3_7_anchor.py")), device=dev)
    batches = bench.make_batches(4, 2, 0, 10, dev, None, True)
else:
    mf = torch.channels_last
    runner = Runner(bench.s2anet_cfg(), device=dev, memory_format=mf, amp_dtype=torch.bfloat16 if dt == "bf16" else None,
                    bf16_params=(dt == "bf16"))
    batches = bench.make_batches(4, 4, 0, 15, dev, mf, False)
for b in batches:
    runner.train_step(*b)
torch.cuda.synchronize()
cnt = collections.Counter()


def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "rs_detection_amd" in f.filename or f.filename.endswith("bench.py")]
    where = "%s:%d %s" % (st[-1].filename.split(ROOT + "/")[-1], st[-1].lineno, st[-1].name) if st else "(the mode's own notice)"
    cnt[where] += 1


warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
runner.train_step(*batches[1])
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
for where, c in cnt.most_common(40):
    print("%3d  %s" % (c, where))
