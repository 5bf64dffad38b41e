"""Root cause of the non-finite loss of a bf16 step whose backbone + neck run as hipGraphs
(torch.cuda.make_graphed_callables): graph the trunk (or the backbone / the neck alone), replay it, and compare every
parameter gradient with eager mode after perturbing the weights.  Finding (MI355X, ROCm 7.2, MIOpen of this image):
the first replay equals eager bit for bit; from the second replay on 8-11 of the 142 weight gradients -- always 1x1
convolutions, a different set every replay -- come back non-finite, while the data gradients stay correct: MIOpen's
bf16 NHWC backward-weight path is not replay-safe (fp32 NCHW is).  And the graphed trunk is SLOWER than eager now that
the step is GPU-bound (25.8 vs 23.3 ms/step), so the capture was dropped rather than worked around.
usage: python profiles/scripts/graph_trunk_bisect.py both|backbone|neck"""
import sys, os, torch, numpy as np
sys.path.insert(0, ".")
import bench
from rs_detection_amd.runner.runner import Runner
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cfg = bench.s2anet_cfg()
r = Runner(cfg, device=dev, memory_format=torch.channels_last, amp_dtype=torch.bfloat16)
m = r.model; m.train()
g = torch.Generator(device="cpu").manual_seed(0)
images = torch.randn(4, 3, 1024, 1024, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
which = sys.argv[1]
class Mod(torch.nn.Module):
    def __init__(s):
        super().__init__()
        if which != 'neck': s.b = m.backbone
        if which != 'backbone': s.n = m.neck
    def forward(s, *x):
        if which == "neck": return tuple(s.n(list(x)))
        if which == "backbone": return tuple(s.b(x[0]))
        return tuple(s.n(s.b(x[0])))
mod = Mod()
with torch.autocast("cuda", dtype=torch.bfloat16):
    feats = [f.detach().clone().requires_grad_(True) for f in m.backbone(images)]
inp = tuple(feats) if which == "neck" else (images,)
with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
    graphed = torch.cuda.make_graphed_callables(mod, tuple(t.detach().clone().requires_grad_(t.requires_grad) for t in inp), allow_unused_input=True)
params = [p for p in mod.parameters() if p.requires_grad]
names = [n for n, p in mod.named_parameters() if p.requires_grad]
def grads(fn, gos=None):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = fn(*inp)
    if gos is None: gos = [torch.randn_like(a) for a in out]
    return torch.autograd.grad(out, params, gos, allow_unused=True), gos
ge, gos = grads(mod)
ge2, _ = grads(mod, gos)
for rep in range(3):
    gg, _ = grads(graphed, gos)
    worst = (0, None); worst2 = (0, None); cnt = 0; bad = []
    for n, a, b, c in zip(names, gg, ge, ge2):
        if a is None or b is None: continue
        cnt += 1
        if not bool(torch.isfinite(a).all()): bad.append(n)
        den = float(b.abs().max()) + 1e-12
        e = float((a - b).abs().max()) / den; e2 = float((c - b).abs().max()) / den
        if e > worst[0]: worst = (e, n)
        if e2 > worst2[0]: worst2 = (e2, n)
    print("compared", cnt, "nonfinite in graph grads:", len(bad), bad)
    print(which, "replay", rep, "graph-vs-eager worst %.3e at %s | eager-vs-eager worst %.3e at %s" % (worst + worst2))
    # perturb weights like an optimizer step
    with torch.no_grad():
        for p in params: p.add_(torch.randn_like(p) * 1e-3 * p.abs().mean())
    ge, _ = grads(mod, gos); ge2, _ = grads(mod, gos)
