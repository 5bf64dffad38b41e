"""Round 6 retry of the captured trunk: backbone + neck of the bf16 S2ANet step as two hipGraphs (forward /
backward) through torch.cuda.make_graphed_callables, inside the running process.  The prepared weight operands of the
one-node Bottleneck (ops/weight_prep.py) are refreshed EAGERLY once per step (their buffers are persistent: the captured
kernels read them in place).  Gate: loss and every gradient of steps 1 and 10 bit-equal to the eager step from the same
weights on the same batch; then eager vs graphed ms/step and host ms/step over `steps` steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from rs_detection_amd.runner.runner import Runner
from rs_detection_amd.ops import weight_prep as wprep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)


def make():
    torch.manual_seed(0)
    r = Runner(bench.s2anet_cfg(), device=dev, memory_format=torch.channels_last, amp_dtype=torch.bfloat16, bf16_params=True)
    r.scheduler = None
    return r


batches = bench.make_batches(1, 4, 0, 15, dev, torch.channels_last, False)
images, targets = batches[0]


def refresh_prepared():
    for reg in wprep._REGISTRIES.values():
        if reg.epoch != wprep._EPOCH[0]:
            reg.refresh()


def run(r, n, graphed):
    losses, grads = [], {}
    for i in range(n):
        if graphed:
            refresh_prepared()
        loss, _ = r.train_step(images, targets)
        losses.append(loss.detach().clone())
        if i in (0, n - 1):
            grads[i] = [None if p.grad is None else p.grad.detach().clone() for p in r.model.parameters()]
    torch.cuda.synchronize()
    return [float(l) for l in losses], grads


def timed(r, n, graphed):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if graphed:
            refresh_prepared()
        r.train_step(images, targets)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3


# ---- eager reference
r = make()
for _ in range(4):
    r.train_step(images, targets)
sd0 = {k: v.detach().clone() for k, v in r.model.state_dict().items()}
ref_losses, ref_grads = run(r, 10, False)
ms_e, host_e = timed(r, steps, False)
print("eager  : %.2f ms/step, host %.2f ms/step" % (ms_e, host_e), flush=True)
del r
torch.cuda.empty_cache()
# ---- graphed trunk, from the same weights (a fresh optimizer: momenta start at zero in both runs' 10 compared steps?  no:
#      the eager run's first 4 steps built momenta -- so the graphed run repeats them eagerly before the capture)
r = make()
for _ in range(4):
    r.train_step(images, targets)
m = r.model


class Trunk(torch.nn.Module):
    def __init__(self, b, n):
        super().__init__()
        self.b, self.n = b, n

    def forward(self, x):
        return tuple(self.n(self.b(x)))


trunk = Trunk(m.backbone, m.neck)
m.train()
sample = (images.contiguous(memory_format=torch.channels_last),)
import gc
r.optimizer.zero_grad(set_to_none=True)
gc.collect()
print("capturing", flush=True)
with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
    g = torch.cuda.make_graphed_callables(trunk, sample, num_warmup_iters=3)
print("captured", flush=True)
orig_forward = m.forward


def fwd(imgs, tg):
    return m.bbox_head(list(g(imgs)), tg)


m.forward = fwd
with torch.no_grad():
    for k, v in m.state_dict().items():
        v.copy_(sd0[k])
wprep.bump_epoch()
# (the optimizer state after 4 eager steps is the same in both runs: same seed, same batch, same arithmetic)
losses, grads = run(r, 10, True)
bad_l = sum(1 for a, b in zip(losses, ref_losses) if a != b)
for i in (0, 9):
    bad = sum(1 for a, b in zip(grads[i], ref_grads[i]) if (a is None) != (b is None) or (a is not None and not torch.equal(a, b)))
    nonfinite = sum(1 for a in grads[i] if a is not None and not torch.isfinite(a.float()).all())
    print("step %2d: loss %.6f (eager %.6f); gradients that differ: %d of %d, non-finite: %d" % (
        i + 1, losses[i], ref_losses[i], bad, len(grads[i]), nonfinite))
print("losses that differ over 10 steps: %d" % bad_l)
ms_g, host_g = timed(r, steps, True)
print("graphed: %.2f ms/step, host %.2f ms/step (eager %.2f / %.2f)" % (ms_g, host_g, ms_e, host_e))
