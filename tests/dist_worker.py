"""One rank of the 2-rank data-parallel check (started by tests/test_gpu_dist.py through
rs_detection_amd.utils.dist.launch_ranks; not a test module itself).

Every rank: the REAL model (S2ANet-R50-FPN or Oriented R-CNN) under DDP on its own shard of a synthetic batch ->
  * gradients after one backward == mean over the shards of the gradients a single, un-wrapped copy of the model
    produces (the reference's semantics: Jittor's optimizer all-reduces grads with op "mean", optimizer.py:30-31);
  * after two optimizer steps every rank holds bit-identical parameters.
Results go to <out>.rank<r>.json; rank 0's file carries the cross-rank comparison."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def make_batch(model_name, rank, dev, size, n=2):
    from rs_detection_amd.utils import synthetic as syn
    g = torch.Generator().manual_seed(50 + rank)
    images = torch.randn(n, 3, size, size, generator=g).to(dev)
    targets = []
    ncls = 10 if model_name == "orcnn" else 15
    for t in syn.synthetic_targets(n, rank=rank, it=0, img=size, num_classes=ncls):
        t = dict(t)
        t["rboxes"] = torch.from_numpy(t["rboxes"][:12]).to(dev)
        t["labels"] = torch.from_numpy(t["labels"][:12]).to(dev)
        if model_name == "orcnn":
            t["hboxes"] = None
        targets.append(t)
    return images, targets


def build_runner(model_name, dev, amp, distributed, channels_last=False):
    from rs_detection_amd.config import Config
    from rs_detection_amd.runner.runner import Runner
    if model_name == "orcnn":
        cfg = Config(os.path.join(ROOT, "configs", "orcnn", "orcnn_van3_7_anchor.py"))
        cfg.model["backbone"] = dict(type="van_b0", img_size=256, num_stages=4, out_indices=(0, 1, 2, 3))
        cfg.model["neck"]["in_channels"] = [32, 64, 160, 256]
    else:
        cfg = Config(os.path.join(ROOT, "configs", "s2anet", "s2anet_r50_fpn_1x_dota.py"))
    torch.manual_seed(0)                       # same initial weights on every rank
    return Runner(cfg, device=dev, distributed=distributed, amp_dtype=amp,
                  memory_format=torch.channels_last if channels_last else None)


def flat_grads(model):
    return torch.cat([p.grad.detach().float().reshape(-1) for p in model.parameters() if p.requires_grad and p.grad is not None])


def main():
    model_name, dtype, out = sys.argv[1], sys.argv[2], sys.argv[3]
    size = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    flags = sys.argv[5].split("+") if len(sys.argv) > 5 else []
    force = "force" in flags          # a ONE-rank group with the reducer forced on (RCCL on one GPU)
    torch_ddp = "ddp" in flags        # torch's DistributedDataParallel instead of utils/reducer.GradReducer
    from rs_detection_amd.utils import dist as rdist
    from rs_detection_amd.utils.general import parse_losses
    rank, local_rank, world = rdist.init_distributed(force=force)
    assert torch.cuda.is_available()
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    amp = torch.bfloat16 if dtype == "bf16" else None
    # "f32cl": the fp32 step in channels_last (the bench's layout): canvas head, 1x1 GEMM split, FusedSGD under DDP
    mode = ("ddp-force" if force else "ddp") if torch_ddp else ("force" if force else True)
    runner = build_runner(model_name, dev, amp, distributed=mode, channels_last=dtype == "f32cl")
    if torch_ddp:
        assert runner.ddp is not runner.model and runner.reducer is None, "DDP wrapper missing"
    else:
        assert runner.reducer is not None and runner.ddp is runner.model, "gradient reducer missing"
    assert (runner.grad_dtype == torch.bfloat16) == (dtype == "bf16")
    reducer = runner.reducer

    def fwd_bwd(module, images, targets, seed, reduce=False):
        torch.manual_seed(seed)                # the RoI / RPN samplers draw from torch's generator
        module.zero_grad(set_to_none=True)
        if amp is not None:
            with torch.autocast("cuda", dtype=amp):
                losses = module(images, targets)
        else:
            losses = module(images, targets)
        total, _ = parse_losses(losses)
        total.backward()
        if reduce and reducer is not None:
            reducer.reduce()                   # what Runner.train_step does between backward() and the optimizer step
        return float(total.detach())

    # -- single-process reference: an un-wrapped copy on every shard, averaged
    ref_model = copy.deepcopy(runner.model)
    ref_model.train()
    runner.model.train()
    def reference():
        acc = None
        for r in range(world):
            im, tg = make_batch(model_name, r, dev, size)
            fwd_bwd(ref_model, im, tg, 100 + r)
            g = flat_grads(ref_model)
            acc = g if acc is None else acc + g
        return acc / world

    want = reference()
    # run-to-run noise floor of the single-process computation itself (atomics in MIOpen's bf16 weight-gradient
    # kernels, amplified by a randomly initialised 50-layer trunk): the DDP comparison cannot be tighter than this
    noise = float((reference() - want).norm() / want.norm().clamp_min(1e-12))
    # -- DDP on this rank's shard
    images, targets = make_batch(model_name, rank, dev, size)
    fwd_bwd(ref_model, images, targets, 100 + rank)
    own = flat_grads(ref_model).clone()
    with (runner.ddp if torch_ddp else reducer).no_sync():   # diagnostic: LOCAL gradients == the plain module's
        fwd_bwd(runner.ddp, images, targets, 100 + rank, reduce=True)
    local = flat_grads(runner.model)
    local_rel = float((local - own).norm() / own.norm().clamp_min(1e-12))
    loss = fwd_bwd(runner.ddp, images, targets, 100 + rank, reduce=True)
    got = flat_grads(runner.model)
    rel = float((got - want).norm() / want.norm().clamp_min(1e-12))
    maxabs = float((got - want).abs().max())
    # -- two real steps, then every rank must hold the same parameters
    for it in range(2):
        torch.manual_seed(200 + 10 * it + rank)
        total, _ = runner.train_step(images, targets)
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().float().reshape(-1) for p in runner.model.parameters()])
    lo, hi = flat.clone(), flat.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    spread = float((hi - lo).abs().max())
    res = dict(rank=rank, world=world, backend=dist.get_backend(), loss=loss, grad_rel_err=rel, grad_max_abs=maxabs,
               grad_norm=float(want.norm()), noise=noise, local_rel_err=local_rel, param_spread=spread, final_loss=float(total.detach()),
               n_grad=int(got.numel()), finite=bool(torch.isfinite(flat).all()),
               sync_mean=rdist.sync_mean(dict(a=torch.tensor(1.0 + rank), b=torch.tensor(3.0)), dev),
               comm_hook=(runner.ddp._comm_hooks[0][0].__qualname__ if getattr(runner.ddp, "_comm_hooks", None) else None),
               reducer="torch_ddp" if torch_ddp else "GradReducer",
               wire=None if reducer is None else reducer.wire_dtypes,
               n_buckets=None if reducer is None else len(reducer.buckets),
               grads_in_buckets=None if reducer is None else all(
                   reducer.owns(p.grad) for p in runner.model.parameters() if p.requires_grad and p.grad is not None),
               bf16_params=bool(runner.bf16_params), optimizer=type(runner.optimizer).__name__,
               bucket_view=bool(getattr(runner.ddp, "gradient_as_bucket_view", False)) if torch_ddp else True)
    with open("%s.rank%d.json" % (out, rank), "w") as f:
        json.dump(res, f)
    rdist.barrier()
    rdist.shutdown()


if __name__ == "__main__":
    main()
