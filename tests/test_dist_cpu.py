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


def test_launch_ranks_relays_rank0_and_reports_the_failing_rank():
    from rs_detection_amd.utils import dist as rdist
    rc, out = rdist.launch_ranks(2, ["-c", "import os; print('rank', os.environ['RANK'], 'of', os.environ['WORLD_SIZE'])"])
    assert rc == 0 and out.strip() == "rank 0 of 2"          # only rank 0's stdout is relayed
    # a rank that dies takes the job down with ITS exit code; the survivor (stuck in a 'collective') is ended
    code = "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(7)\ntime.sleep(60)"
    rc, _ = rdist.launch_ranks(2, ["-c", code], timeout=30)
    assert rc == 7


@pytest.mark.timeout(300)
def test_bench_gpus2_self_launches_and_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` with no launcher around it: the parent must start the ranks itself (round 1 died on
    an assert here).  On this GPU-less container every rank then refuses to run (no CPU fallback) -> non-zero exit."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=280)
    if torch.cuda.device_count() > 0:
        pytest.skip("GPU present: covered by tests/test_gpu_dist.py")
    assert p.returncode != 0 and "AssertionError" not in p.stderr
    assert "needs an MI355X" in p.stderr


def _bf16_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from rs_detection_amd.utils import dist as rdist
    rdist.init_distributed(backend="gloo")
    torch.manual_seed(0)
    model = torch.nn.Linear(64, 64)
    ddp = rdist.wrap_ddp(model, torch.device("cpu"), bucket_cap_mb=1, grad_dtype=torch.bfloat16)
    x = torch.randn(8, 64, generator=torch.Generator().manual_seed(rank))
    ddp(x).square().mean().backward()
    g = model.weight.grad.clone()
    # reference: fp32 mean of both ranks' gradients
    ref = torch.nn.Linear(64, 64)
    ref.load_state_dict(model.state_dict())
    tot = 0
    for r in range(world):
        ref.zero_grad()
        ref(torch.randn(8, 64, generator=torch.Generator().manual_seed(r))).square().mean().backward()
        tot = tot + ref.weight.grad
    q.put((rank, g.dtype == torch.float32, float((g - tot / world).abs().max() / (tot / world).abs().max()),
           bool((g == g.bfloat16().float()).all())))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_bf16_gradient_buckets():
    """configs[2..4]: buckets cross the wire in bf16 (half the bytes per xGMI link), gradients come back fp32."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bf16_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, is_f32, rel, on_grid in out:
        assert is_f32 and rel < 2e-2          # bf16 has 8 significant bits
        assert on_grid                        # every value is a bf16 number: the hook really compressed the bucket


# ---- 8 ranks without the hardware: affinity, the shared MIOpen record directory, an 8-rank smoke -----
def test_pin_rank_to_cores_splits_the_allowed_cores():
    import subprocess
    code = ("import os, sys, json\nsys.path.insert(0, %r)\nfrom rs_detection_amd.utils import dist as d\n"
            "before = sorted(os.sched_getaffinity(0))\nmine = d.pin_rank_to_cores()\n"
            "print(json.dumps([before, mine, sorted(os.sched_getaffinity(0))]))" % ROOT)
    import json
    allowed = sorted(os.sched_getaffinity(0))
    n = 4 if len(allowed) >= 4 else len(allowed)
    if n < 2:
        pytest.skip("one core")
    shares = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
        before, mine, after = json.loads(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True,
                                                        text=True, check=True).stdout.strip().splitlines()[-1])
        assert mine == after and len(mine) == len(allowed) // n and set(mine) <= set(before)
        shares.append(set(mine))
    assert all(shares[i].isdisjoint(shares[j]) for i in range(n) for j in range(i))       # no two ranks share a core
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), RSDET_NO_AFFINITY="1")
    before, mine, after = json.loads(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True,
                                                    text=True, check=True).stdout.strip().splitlines()[-1])
    assert mine is None and after == before


def _fake_sysfs(root, gpu_nodes, node_cpus, siblings):
    """A minimal /sys: drm cards (AMD, PCI slot names in order) with their NUMA node, node cpulists, SMT siblings."""
    for i, node in enumerate(gpu_nodes):
        d = os.path.join(root, "class", "drm", "card%d" % (7 - i), "device")        # card numbers need not follow PCI order
        os.makedirs(d)
        for name, text in (("vendor", "0x1002"), ("numa_node", str(node)), ("uevent", "DRIVER=amdgpu\nPCI_SLOT_NAME=0000:%02x:00.0" % (0x10 + i))):
            with open(os.path.join(d, name), "w") as f:
                f.write(text + "\n")
    os.makedirs(os.path.join(root, "class", "drm", "card0-DP-1", "device"))      # a connector: not a device
    for n, cpus in node_cpus.items():
        os.makedirs(os.path.join(root, "devices", "system", "node", "node%d" % n))
        with open(os.path.join(root, "devices", "system", "node", "node%d" % n, "cpulist"), "w") as f:
            f.write(cpus + "\n")
    for c, sib in siblings.items():
        d = os.path.join(root, "devices", "system", "cpu", "cpu%d" % c, "topology")
        os.makedirs(d)
        with open(os.path.join(d, "thread_siblings_list"), "w") as f:
            f.write(sib + "\n")


def test_pin_rank_to_cores_follows_the_gpu_numa_node(tmp_path, monkeypatch):
    """8 GPUs on 2 sockets (4 + 4), 16 physical cores with SMT (cpu c and c + 16 are siblings), socket 0 = physical
    cores 0-7, socket 1 = 8-15: a rank gets whole physical cores of ITS GPU's socket; ranks of one socket split it."""
    from rs_detection_amd.utils import dist as d
    root = str(tmp_path / "sys")
    _fake_sysfs(root, gpu_nodes=[0, 0, 1, 1, 0, 0, 1, 1], node_cpus={0: "0-7,16-23", 1: "8-15,24-31"},
                siblings={c: "%d,%d" % (c % 16, c % 16 + 16) for c in range(32)})
    assert d.gpu_numa_nodes(root) == [0, 0, 1, 1, 0, 0, 1, 1]
    applied = {}
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(32)))
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cores: applied.__setitem__("cores", sorted(cores)))
    monkeypatch.setattr(d.torch, "set_num_threads", lambda n: None)
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "RSDET_NO_AFFINITY"):
        monkeypatch.delenv(var, raising=False)
    shares = []
    for r in range(8):
        mine = d.pin_rank_to_cores(local_rank=r, local_world=8, sysfs=root)
        assert mine == applied["cores"] and len(mine) == 4                      # 2 physical cores x 2 threads
        node_cpus = set(range(0, 8)) | set(range(16, 24)) if r in (0, 1, 4, 5) else set(range(8, 16)) | set(range(24, 32))
        assert set(mine) <= node_cpus, (r, mine)                                # on the socket of its own GPU
        assert all((c + 16) % 32 in mine for c in mine)                         # SMT siblings stay together
        shares.append(set(mine))
    assert all(shares[i].isdisjoint(shares[j]) for i in range(8) for j in range(i))
    # HIP_VISIBLE_DEVICES remaps local ranks to physical GPUs: rank 0 -> GPU 2 (socket 1), rank 1 -> GPU 0 (socket 0)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,0")
    assert set(d.pin_rank_to_cores(local_rank=0, local_world=2, sysfs=root)) <= set(range(8, 16)) | set(range(24, 32))
    assert set(d.pin_rank_to_cores(local_rank=1, local_world=2, sysfs=root)) <= set(range(0, 8)) | set(range(16, 24))
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    # a restricted process (cgroup of 8 CPUs, all on socket 0) with GPUs on both sockets: the socket-1 ranks find none
    # of their socket's cores allowed, so the allowed cores are split evenly instead of leaving ranks without a share
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(0, 4)) | set(range(16, 20)))
    got = [d.pin_rank_to_cores(local_rank=r, local_world=4, sysfs=root) for r in range(4)]
    assert all(g and len(g) == 2 for g in got) and len(set(map(tuple, got))) == 4
    # no sysfs at all (a container that hides it): even split by local rank
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(8)))
    got = [d.pin_rank_to_cores(local_rank=r, local_world=4, sysfs=str(tmp_path / "nothing")) for r in range(4)]
    assert got == [[0, 1], [2, 3], [4, 5], [6, 7]]


def _db_racer(tmp, q):
    os.environ["XDG_CACHE_HOME"] = tmp
    os.environ.pop("MIOPEN_USER_DB_PATH", None)
    os.environ.pop("RSDET_MIOPEN_DB_IN_USE", None)
    os.environ.pop("RSDET_NO_MIOPEN_DB", None)
    sys.path.insert(0, ROOT)
    from rs_detection_amd.utils.miopen_db import use_packaged_miopen_db
    d = use_packaged_miopen_db()
    # (the "<file>.src" stamps name the packaged file a working copy came from: not record files)
    q.put((d, sorted((f, os.path.getsize(os.path.join(d, f))) for f in os.listdir(d) if not f.endswith(".src")) if d else None))


@pytest.mark.timeout(180)
def test_eight_processes_race_on_one_miopen_db_directory(tmp_path):
    """The 8 ranks of a node start together and copy the packaged MIOpen records into ONE per-user directory: every
    rank must end up with complete files (atomic replace), no temporaries left behind."""
    import glob
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_db_racer, args=(str(tmp_path), q)) for _ in range(8)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    pkg = os.path.join(ROOT, "rs_detection_amd", "miopen_db")
    want = sorted((os.path.basename(f), os.path.getsize(f)) for f in glob.glob(os.path.join(pkg, "*db.txt")))
    assert want, "packaged records missing"
    for d, files in outs:
        assert d == outs[0][0] and files == want, (d, files)                    # same directory, whole files, no *.tmp


def _rank8(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import numpy as np
    import torch.distributed as dist
    from rs_detection_amd.utils import dist as rdist
    from rs_detection_amd.data import SyntheticDOTADataset
    r, _, w = rdist.init_distributed(backend="gloo")
    ds = SyntheticDOTADataset(tile=32, batch_size=2, num_images=37, shuffle=True)
    ds.set_shard(r, w)                                   # training split: whole global batches, equal steps per rank
    train = [t["filename"] for _, tg in ds for t in tg]
    ds.set_shard(r, w, keep_all=True)                    # evaluation split: every image exactly once
    val = [t["filename"] for _, tg in ds for t in tg]
    allval = [n for part in rdist.gather_objects(val) for n in part]
    synced = rdist.sync_mean({"loss": torch.tensor(float(r))}, torch.device("cpu"))
    tmax = rdist.all_reduce_max(float(r), torch.device("cpu"))
    q.put((r, len(train), sorted(train), len(set(allval)), len(allval), synced["loss"], tmax,
           sorted(os.sched_getaffinity(0))))
    rdist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_eight_rank_gloo_smoke_shards_sync_and_gather():
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [o[0] for o in out] == list(range(8))
    assert len({o[1] for o in out}) == 1 and out[0][1] == 4                     # 37 images, 8 ranks x batch 2: 2 steps each
    seen = [n for o in out for n in o[2]]
    assert len(seen) == len(set(seen)) == 32                                     # disjoint training shards
    assert all(o[3] == 37 and o[4] == 37 for o in out)                           # sharded validation covers every image once
    assert all(abs(o[5] - 3.5) < 1e-6 and o[6] == 7.0 for o in out)              # mean / max over 8 ranks
    cores = [set(o[7]) for o in out]
    if len(os.sched_getaffinity(0)) >= 8:                                        # init_distributed pinned every rank
        assert all(cores[i].isdisjoint(cores[j]) for i in range(8) for j in range(i))


def _force_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from rs_detection_amd.utils import dist as rdist
    assert rdist.init_distributed(backend="gloo") == (0, 0, 1) and not dist.is_initialized()   # world 1: no group ...
    rdist.init_distributed(backend="gloo", force=True)                                         # ... unless forced
    assert dist.is_initialized() and dist.get_world_size() == 1
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 1))
    assert rdist.wrap_ddp(model, torch.device("cpu")) is model
    ddp = rdist.wrap_ddp(model, torch.device("cpu"), bucket_cap_mb=1, grad_dtype=torch.bfloat16, force=True)
    assert ddp is not model and ddp._comm_hooks
    x = torch.randn(2, 3, 8, 8)
    ddp(x).square().mean().backward()
    got = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    model(x).square().mean().backward()
    ok = all(torch.allclose(a, p.grad, rtol=2e-2, atol=1e-4) for a, p in zip(got, model.parameters()))   # bf16 wire
    q.put(bool(ok))
    rdist.shutdown()


def test_forced_one_rank_group_runs_the_reducer_path():
    """wrap_ddp(force=True) in a one-rank process group: the reducer (buckets, compress hook, all-reduce) runs although
    there is nobody to reduce with -- the form tests/test_gpu_dist.py uses to put the step through RCCL on one GPU."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_force_worker, args=(0, 1, _free_port(), q))
    p.start()
    ok = q.get(timeout=120)
    p.join(60)
    assert ok and p.exitcode == 0


def test_visible_gpu_does_not_guess_through_composed_masks(monkeypatch):
    from rs_detection_amd.utils import dist as rdist
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    assert rdist._visible_gpu(3) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5,6,7")
    assert rdist._visible_gpu(1) == 5
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0,1,2,3,4,5,6,7")
    assert rdist._visible_gpu(1) is None            # ROCR and HIP masks compose: do not re-derive, take the even split
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    assert rdist._visible_gpu(1) == 1


def test_check_pinning_reports_the_node_of_the_opened_gpu(tmp_path, monkeypatch):
    from rs_detection_amd.utils import dist as rdist
    root = tmp_path / "sys"
    (root / "bus" / "pci" / "devices" / "0000:c1:00.0").mkdir(parents=True)
    (root / "bus" / "pci" / "devices" / "0000:c1:00.0" / "numa_node").write_text("1\n")
    (root / "devices" / "system" / "node" / "node1").mkdir(parents=True)
    (root / "devices" / "system" / "node" / "node1" / "cpulist").write_text("8-15\n")

    class P:
        pci_domain_id, pci_bus_id, pci_device_id = 0, 0xc1, 0
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda i: P)
    said = []
    assert rdist.check_pinning(0, [8, 9, 10], sysfs=str(root), log=said.append) == (1, True)
    assert rdist.check_pinning(0, [0, 1], sysfs=str(root), log=said.append) == (1, False)
    assert "NOT ALL ON" in said[-1]


def _reducer_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from rs_detection_amd.utils import dist as rdist
    from rs_detection_amd.utils.reducer import GradReducer
    rdist.init_distributed(backend="gloo")
    torch.manual_seed(rank)                      # different weights per rank: the reducer's broadcast must fix that
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 2, 1))
    model = model.to(memory_format=torch.channels_last)
    model[2].bias.requires_grad_(False)          # a frozen parameter is not part of any bucket
    red = GradReducer(model, bucket_cap_mb=0.0001)
    params = [p for p in model.parameters() if p.requires_grad]
    x = torch.randn(2, 3, 8, 8, generator=torch.Generator().manual_seed(100 + rank))

    launched_early = []

    def backward(protocol=True, keep=False):
        if not keep:
            for p in model.parameters():
                p.grad = None
        if protocol:
            red.begin_step()                     # (what Runner.train_step does between zero_grad and backward)
        (model(x).square().mean() * (rank + 1)).backward()
        launched_early.append(red._next)         # buckets the hooks sent from inside backward
        red.reduce()
        return torch.cat([p.grad.reshape(-1) for p in params])
    backward()                                   # first step: records (and cross-checks) the used set, nothing leaves early
    got = backward()                             # second step: armed -- every bucket leaves from inside backward
    early = list(launched_early)
    unarmed = backward(protocol=False)           # a caller without begin_step: same result, everything sent by reduce()
    # gradients zeroed in place (zero_grad(set_to_none=False)): p.grad IS the bucket view when backward starts -- the hooks
    # must NOT take "present" for "produced" (not armed), the sum lands in the bucket, the mean is the same
    for p in params:
        p.grad.zero_()
    inplace = backward(keep=True)
    same = float(max((unarmed - got).abs().max(), (inplace - got).abs().max()))
    # accumulation: a local step under no_sync, then a synchronised one on top of it == mean over ranks of 2 x local
    with red.no_sync():
        backward()
    accum = backward(keep=True)
    owned = all(red.owns(p.grad) for p in params) and all(p.grad.stride() == p.stride() for p in params)
    with red.no_sync():
        local = backward()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    ws = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(ws, w)
    mean = sum(gathered) / world
    q.put((rank, float((got - mean).abs().max()), float((ws[0] - ws[1]).abs().max()), len(red.buckets),
           owned, not any(red.owns(p.grad) for p in params),       # (under no_sync the gradients stay local tensors)
           early, launched_early[2:4], same, float((accum - 2 * mean).abs().max())))
    rdist.shutdown()


def test_grad_reducer_two_ranks_gloo():
    """utils/reducer.GradReducer: bucketed mean of the gradients over two ranks == the mean of the local gradients
    (optims/optimizer.py:30-31 of the reference: all-reduce with op "mean"); every rank starts from rank 0's weights;
    p.grad ends up a view into a bucket with the parameter's own strides (channels_last included)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=180) for _ in range(2))
    [p.join(60) for p in ps]
    for rank, err, spread, nb, owned, strides, early, late, same, accum in res:
        assert err <= 1e-7 and spread == 0.0 and nb == 2 and owned and strides, res
        assert early == [0, 2], res             # step 1 records the used set; step 2 (armed) sends both buckets from the hooks
        assert late == [0, 0], res              # no begin_step / gradients present at the start: nothing leaves early
        assert same <= 1e-7 and accum <= 1e-6, res


def _reducer_mismatch_worker(rank, world, port, q):
    """rank 1 leaves the middle layer out of its graph.  Round 5's reducer returned OK with every gradient
    averaged against the wrong bucket; now every rank raises at the first reduce()."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from rs_detection_amd.utils import dist as rdist
    from rs_detection_amd.utils.reducer import GradReducer
    rdist.init_distributed(backend="gloo")
    torch.manual_seed(0)
    layers = torch.nn.ModuleList([torch.nn.Linear(8, 8) for _ in range(3)])
    red = GradReducer(layers, bucket_cap_mb=0.0001)
    x = torch.randn(4, 8)

    def step(skip_middle):
        for p in layers.parameters():
            p.grad = None
        red.begin_step()
        h = layers[0](x)
        if not skip_middle:
            h = layers[1](h)
        layers[2](h).square().mean().backward()
        red.reduce()
    msg = None
    try:
        step(skip_middle=(rank == 1))
    except RuntimeError as e:
        msg = str(e)
    q.put((rank, "first", msg))
    rdist.shutdown()


def _reducer_drift_worker(rank, world, port, q):
    """... and a used set that changes AFTER the first step is refused locally, before any collective of that step."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from rs_detection_amd.utils import dist as rdist
    from rs_detection_amd.utils.reducer import GradReducer
    rdist.init_distributed(backend="gloo", force=True)
    torch.manual_seed(0)
    layers = torch.nn.ModuleList([torch.nn.Linear(8, 8) for _ in range(3)])
    red = GradReducer(layers, bucket_cap_mb=0.0001)
    x = torch.randn(4, 8)

    def step(skip_middle):
        for p in layers.parameters():
            p.grad = None
        red.begin_step()
        h = layers[0](x)
        if not skip_middle:
            h = layers[1](h)
        layers[2](h).square().mean().backward()
        red.reduce()
    step(True)
    unused_stay_none = layers[1].weight.grad is None and layers[0].weight.grad is not None
    step(True)
    msg = None
    try:
        step(False)
    except RuntimeError as e:
        msg = str(e)
    q.put((unused_stay_none, msg))
    rdist.shutdown()


def test_grad_reducer_refuses_rank_dependent_and_drifting_graphs():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_reducer_mismatch_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=180) for _ in range(2))
    [p.join(60) for p in ps]
    for rank, _, msg in res:                    # BOTH ranks raise (nobody is left waiting in a collective)
        assert msg is not None and "ranks disagree" in msg and "1.weight" in msg, res
    p = ctx.Process(target=_reducer_drift_worker, args=(0, 1, _free_port(), q))
    p.start()
    unused_stay_none, msg = q.get(timeout=120)
    p.join(60)
    assert unused_stay_none                     # a parameter no rank uses keeps grad None (no decay / momentum on it)
    assert msg is not None and "must not change between steps" in msg and "1." in msg


def test_rank_facts_and_rccl_requirement(monkeypatch):
    """bench.py's ``multi_gpu`` object is built from rank_facts(); a job with a GPU per rank that is NOT on RCCL is refused."""
    from rs_detection_amd.utils import dist as rdist
    f = rdist.rank_facts(torch.device("cpu"))
    assert f["rank"] == 0 and f["group_world"] == 1 and f["backend"] is None and f["device"] is None
    rdist.require_rccl(1)                                   # one rank: nothing to require
    import torch.distributed as dist
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_backend", lambda *a: "gloo")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    with pytest.raises(SystemExit, match="RCCL"):
        rdist.require_rccl(8)                               # 8 GPUs, 8 ranks, gloo: refused
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    rdist.require_rccl(2)                                   # two ranks SHARING one GPU (the 1-GPU tests): gloo is legitimate
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(dist, "get_backend", lambda *a: "nccl")
    rdist.require_rccl(8)
