"""CPU, world_size 2, gloo: the N>1 path -- rendezvous from torchrun-style env vars, per-rank shards of
the synthetic tile stream, DDP gradient all-reduce (mean), max-over-ranks timing, metric sync."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from rs_detection_amd.utils import dist as rdist
    from rs_detection_amd.utils import synthetic as syn
    r, lr, w = rdist.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    dev = torch.device("cpu")
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 1))
    ddp = rdist.wrap_ddp(model, dev, bucket_cap_mb=1)
    assert ddp is not model
    # each rank draws its own tiles (pure data parallelism, no data-path collective)
    tg = syn.synthetic_targets(2, rank=rank, it=0)
    g = torch.Generator().manual_seed(rank)
    x = torch.randn(2, 3, 8, 8, generator=g)
    loss = ddp(x).square().mean() * (rank + 1)
    loss.backward()
    grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    # reference: same computation on both shards, averaged
    ref_model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 1))
    ref_model.load_state_dict(model.state_dict())
    tot = None
    for rr in range(world):
        gg = torch.Generator().manual_seed(rr)
        xx = torch.randn(2, 3, 8, 8, generator=gg)
        ref_model.zero_grad()
        (ref_model(xx).square().mean() * (rr + 1)).backward()
        gr = torch.cat([p.grad.reshape(-1) for p in ref_model.parameters()])
        tot = gr if tot is None else tot + gr
    ok_grad = torch.allclose(grads, tot / world, atol=1e-6)
    tmax = rdist.all_reduce_max(1.0 + rank, dev)
    synced = rdist.sync_mean({"loss": torch.tensor(float(rank + 1))}, dev)
    rdist.barrier()
    q.put((rank, ok_grad, tmax, synced["loss"], [t["rboxes"].shape[0] for t in tg], float(tg[0]["rboxes"][0, 0])))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_gloo_ddp():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert [o[1] for o in out] == [True, True]                 # all-reduced (mean) gradients
    assert [o[2] for o in out] == [2.0, 2.0]                   # MAX over ranks timing
    assert [o[3] for o in out] == [1.5, 1.5]                   # metric sync = mean
    assert out[0][4] == out[1][4] == [16, 100]                 # same K cycle per rank ...
    assert out[0][5] != out[1][5]                              # ... different tiles


def test_single_process_is_a_noop():
    from rs_detection_amd.utils import dist as rdist
    m = torch.nn.Linear(2, 2)
    assert rdist.wrap_ddp(m, torch.device("cpu")) is m
    assert rdist.all_reduce_max(3.5, torch.device("cpu")) == 3.5
