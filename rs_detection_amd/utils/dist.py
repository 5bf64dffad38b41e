"""One process per GPU over torch.distributed (backend "nccl" == RCCL on ROCm, xGMI intra-node).

Replaces Jittor's MPI launcher + in-optimizer gradient all-reduce
(/root/reference/python/jdet/optims/optimizer.py:30-31 -> jittor Optimizer.pre_step;
metric sync /root/reference/python/jdet/utils/general.py:30-48; launcher ``mpirun -np 8 python tools/run_net.py``,
/root/reference/README_competition.md:79-80).  The path shards by image (pure data parallelism): each rank
draws its own tiles; the only collective is the bucketed gradient all-reduce DDP overlaps with backward.

``launch_ranks`` is the launcher half: it starts one child process per rank BEFORE the calling process has touched
the GPU (a process that has initialised HIP must never be replaced or forked into ranks) and relays rank 0's output.
"""
import datetime
import glob
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _cpulist(text):
    """'0-3,8-11' -> [0, 1, 2, 3, 8, 9, 10, 11]"""
    out = []
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_nodes(sysfs="/sys"):
    """NUMA node of every AMD GPU of the machine in PCI-address order (the order HIP enumerates them in), read from
    ``/sys/class/drm/card*/device/numa_node`` -- no HIP call.  None for a GPU whose node the kernel does not know (-1)."""
    found = {}
    for dev in glob.glob(os.path.join(sysfs, "class", "drm", "card[0-9]*", "device")):
        if "-" in os.path.basename(os.path.dirname(dev)):          # card0-DP-1 ...: connectors, not devices
            continue
        if (_read(os.path.join(dev, "vendor")) or "").lower() != "0x1002":
            continue
        slot = None
        for ln in (_read(os.path.join(dev, "uevent")) or "").splitlines():
            if ln.startswith("PCI_SLOT_NAME="):
                slot = ln.split("=", 1)[1]
        slot = slot or os.path.basename(os.path.realpath(dev))
        node = _read(os.path.join(dev, "numa_node"))
        found[slot] = int(node) if node not in (None, "", "-1") else None
    return [found[k] for k in sorted(found)]


def _visible_gpu(local_rank):
    """Physical index of the GPU this rank will open: the local rank through the *_VISIBLE_DEVICES remapping.  None
    (= "do not guess": the caller falls back to the even split) for UUID lists and when BOTH a ROCR and a HIP / CUDA
    mask are set -- the runtime applies them one after the other and the composition is not ours to re-derive."""
    masks = [var for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if os.environ.get(var)]
    if "ROCR_VISIBLE_DEVICES" in masks and len(masks) > 1:
        return None
    for var in masks:
        v = os.environ.get(var)
        if v:
            ids = [x.strip() for x in v.split(",") if x.strip()]
            if all(x.isdigit() for x in ids) and local_rank < len(ids):
                return int(ids[local_rank])
            return None                 # UUIDs or fewer devices than ranks: do not guess
    return local_rank


def gpu_numa_node_by_pci(pci_addr, sysfs="/sys"):
    """NUMA node of the GPU at PCI address 'dddd:bb:dd.f' (None if unknown)."""
    node = _read(os.path.join(sysfs, "bus", "pci", "devices", pci_addr, "numa_node"))
    return int(node) if node not in (None, "", "-1") else None


def check_pinning(device_index, cores, sysfs=None, log=None):
    """After the device is open: does the socket the rank was pinned to (from the sysfs enumeration order, an
    ASSUMPTION about HIP's device order) match the node of the GPU it really opened (by PCI address, a fact)?  Reports
    through ``log`` (default: stderr) -- once per rank, a wrong guess costs the cross-socket hop silently otherwise.
    Returns (gpu_node, pinned_nodes) or None where the information is not there."""
    import sys
    sysfs = sysfs or os.environ.get("RSDET_SYSFS_ROOT", "/sys")
    try:
        pr = torch.cuda.get_device_properties(device_index)
        addr = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:                    # noqa: BLE001  (older torch: no PCI fields)
        return None
    node = gpu_numa_node_by_pci(addr, sysfs)
    if node is None or not cores:
        return None
    node_cpus = set(_cpulist(_read(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist"))))
    on_node = bool(node_cpus) and set(cores) <= node_cpus
    rank = env_world()[0]
    msg = "rank %d: GPU %d at %s on NUMA node %d, pinned to %d cores %s that node" % (
        rank, device_index, addr, node, len(cores), "ON" if on_node else "NOT ALL ON")
    if not on_node or rank == 0:
        (log or (lambda m: print(m, file=sys.stderr)))(msg)
    return node, on_node


def _sibling_groups(cores, sysfs="/sys"):
    """``cores`` grouped by physical core (SMT siblings together), groups in ascending order of their first CPU."""
    seen, groups = set(), []
    allowed = set(cores)
    for c in sorted(cores):
        if c in seen:
            continue
        sib = _cpulist(_read(os.path.join(sysfs, "devices", "system", "cpu", "cpu%d" % c, "topology",
                                          "thread_siblings_list")))
        grp = sorted(x for x in (sib or [c]) if x in allowed) or [c]
        seen.update(grp)
        groups.append(grp)
    return groups


def pin_rank_to_cores(local_rank=None, local_world=None, sysfs=None):
    """Give this rank its own share of the host cores ON THE SOCKET ITS GPU HANGS OFF: ``os.sched_setaffinity`` over
    the cores this process is allowed to use.  The GPU's NUMA node comes from sysfs (``gpu_numa_nodes``); the ranks whose
    GPUs share a node split that node's cores evenly, whole physical cores at a time (SMT siblings stay together).
    Where the topology is unknown (no sysfs entry, a container that hides it, more ranks than cores on a node) the
    allowed cores are split evenly by LOCAL_RANK instead.  The bf16 step is bound by the host (one Python thread
    enqueues ~800 launches per step): eight ranks that migrate over -- and share -- the same cores lose to each other
    and to their own loader workers, and a rank on the far socket pays the inter-socket hop on every doorbell; pinned,
    every rank keeps its L2 / NUMA locality and its workers (children of this process) inherit the same share.  Must
    run before the first GPU call of the process (HIP's helper threads inherit the mask they are created under).
    No-op for a single rank, when ``RSDET_NO_AFFINITY=1``, or where the OS has no affinity API.
    Returns the list of cores set (or None)."""
    if os.environ.get("RSDET_NO_AFFINITY", "0") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    rank, lr, world = env_world()
    local_rank = lr if local_rank is None else int(local_rank)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world)) if local_world is None else int(local_world)
    if local_world <= 1:
        return None
    sysfs = sysfs or os.environ.get("RSDET_SYSFS_ROOT", "/sys")
    cores = sorted(os.sched_getaffinity(0))
    mine = None
    # ---- NUMA-aware share
    nodes = gpu_numa_nodes(sysfs)
    phys = [_visible_gpu(r) for r in range(local_world)]
    if nodes and all(p is not None and p < len(nodes) and nodes[p] is not None for p in phys):
        # the decision must be the same on every rank (each one computes it alone): NUMA shares only if EVERY node that
        # hosts a rank has at least one allowed physical core per rank, else all ranks take the even split below
        shares = {}
        for node in sorted(set(nodes[p] for p in phys)):
            peers = [r for r in range(local_world) if nodes[phys[r]] == node]
            node_cpus = set(_cpulist(_read(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist"))))
            groups = _sibling_groups([c for c in cores if c in node_cpus], sysfs)
            per = len(groups) // len(peers)
            if per < 1:
                shares = None
                break
            for k, r in enumerate(peers):
                shares[r] = sorted(c for g in groups[k * per:(k + 1) * per] for c in g)
        if shares:
            mine = shares[local_rank]
    # ---- topology unknown: even split of the allowed cores (whole physical cores where sysfs names the siblings)
    if mine is None:
        groups = _sibling_groups(cores, sysfs)
        per = len(groups) // local_world
        if per < 1:                         # more ranks than cores: leave the scheduler alone
            return None
        mine = sorted(c for g in groups[local_rank * per:(local_rank + 1) * per] for c in g)
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    # intra-op thread pools sized for the share (the default is the machine's core count, per rank)
    os.environ.setdefault("OMP_NUM_THREADS", str(min(len(mine), 8)))
    try:
        torch.set_num_threads(min(len(mine), 8))
    except RuntimeError:
        pass
    return mine


def pick_backend():
    """RCCL when every rank has a GPU of its own; gloo when ranks have to share a device (RCCL refuses two ranks on
    one GPU) or there is none.  ``RSDET_DIST_BACKEND`` overrides.  ``device_count`` does not initialise HIP."""
    forced = os.environ.get("RSDET_DIST_BACKEND")
    if forced:
        return forced
    _, _, world = env_world()
    n = torch.cuda.device_count()
    return "nccl" if (n >= world and n > 0) else "gloo"


def init_distributed(backend=None, timeout_s=1800, force=False):
    """Initialise the default process group from torchrun env vars (no-op for world size 1 unless ``force``: a
    one-rank group, which is how the RCCL reducer path is exercised on a single GPU -- tests/test_gpu_dist.py)."""
    rank, local_rank, world = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        cores = pin_rank_to_cores()     # before this process's first GPU call (set_device / NCCL init below)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = pick_backend()
        if backend == "nccl":
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=timeout_s))
        if backend == "nccl" and cores:
            check_pinning(torch.cuda.current_device(), cores)    # the GPU really opened vs the socket pinned to
    return rank, local_rank, world


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def wrap_ddp(model, device, bucket_cap_mb=64, grad_dtype=None, find_unused_parameters=False, static_graph=None,
             force=False):
    """DDP with gradient-as-bucket-view; 64 MB buckets: the 145 MB fp32 gradient set of
    S2ANet-R50 goes out as ~3 large all-reduces (per-link-bound ring over xGMI favours few,
    large messages) that overlap with the backbone backward.

    ``grad_dtype=torch.bfloat16`` (the bf16 configs, BASELINE configs[2..4]): every bucket is rounded to bf16 for the
    wire and widened again before the optimizer sees it (torch's ``bf16_compress_hook``), which halves the bytes per
    xGMI link (72 MB instead of 145 MB per step); the master gradients and the SGD update stay fp32.

    The model's only graph-less parameters (RotationInvariantPooling's unused conv/BN) are frozen
    (``requires_grad=False``), so DDP never waits for them and ``find_unused_parameters`` can stay off.

    ``static_graph`` (default off).  Evaluated in round 3 and left off: the steps use
    the same parameters in the same order every iteration, so it is legal, but with torch 2.10 + ROCm the 2-rank
    harness (tests/dist_worker.py: a second backward through an un-wrapped copy sharing the parameters) trips the
    reducer's internal assert `expect_autograd_hooks_` (reducer.cpp:1703) under static_graph=True; plain DDP passes the
    same harness, and the bookkeeping it would save is ~0.1 ms of a 23-57 ms step.

    ``force``: wrap at world size 1 too (a one-rank process group must exist): the whole reducer path -- bucket views,
    compress hook, the all-reduce through RCCL -- then runs on one GPU, which is how it is tested without a node and how
    ``bench.py`` measures what DDP adds to the host side of a step (``ddp_host_overhead_ms``)."""
    static_graph = bool(static_graph)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return model
    from torch.nn.parallel import DistributedDataParallel as DDP
    ids = [device.index] if device.type == "cuda" else None
    ddp = DDP(model, device_ids=ids, gradient_as_bucket_view=True, bucket_cap_mb=bucket_cap_mb,
              broadcast_buffers=False, find_unused_parameters=find_unused_parameters, static_graph=bool(static_graph))
    if grad_dtype == torch.bfloat16:
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        ddp.register_comm_hook(None, default_hooks.bf16_compress_hook)
    elif grad_dtype == torch.float16:
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        ddp.register_comm_hook(None, default_hooks.fp16_compress_hook)
    return ddp


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def all_reduce_max(value, device):
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sync_mean(values, device):
    """utils/general.py:30-48 ``sync``: mean-all-reduce a dict of scalars, return Python floats."""
    keys = sorted(values)
    t = torch.stack([values[k].detach().float().reshape(()) for k in keys]).to(device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t)
        t /= dist.get_world_size()
    return dict(zip(keys, t.tolist()))


def rank_facts(device=None):
    """What a multi-GPU run should be able to show about THIS rank: the device it opened (index, PCI address, the NUMA node of
    that PCI function), the cores it is pinned to, the collective backend and the group size as the backend itself reports
    it.  Gathered by bench.py into the line's ``multi_gpu`` object."""
    rank, local_rank, world = env_world()
    f = {"rank": rank, "local_rank": local_rank, "env_world": world, "visible_gpus": torch.cuda.device_count(),
         "backend": None, "group_world": 1, "device": None, "pci": None, "numa_node": None,
         "cores": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    if dist.is_available() and dist.is_initialized():
        f["backend"], f["group_world"] = dist.get_backend(), dist.get_world_size()
    if device is not None and getattr(device, "type", None) == "cuda":
        idx = device.index if device.index is not None else torch.cuda.current_device()
        f["device"] = idx
        try:
            pr = torch.cuda.get_device_properties(idx)
            f["pci"] = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            f["numa_node"] = gpu_numa_node_by_pci(f["pci"], os.environ.get("RSDET_SYSFS_ROOT", "/sys"))
        except Exception:                # noqa: BLE001
            pass
    return f


def require_rccl(world):
    """A multi-rank job on a node that HAS a GPU per rank must reduce over RCCL: a silent fall-back to gloo (a stale
    RSDET_DIST_BACKEND, a build without RCCL) would measure host-memory all-reduces.  Raises SystemExit (non-zero exit of
    every rank, from the process that started them -- never a re-exec); ranks SHARING a GPU (the 2-rank tests on a 1-GPU
    box) are the one legitimate gloo case."""
    if world > 1 and dist.is_initialized() and torch.cuda.device_count() >= world and dist.get_backend() != "nccl":
        raise SystemExit("rs_detection_amd: %d ranks on a node with %d visible GPUs, but the process group runs on %r -- the "
                         "gradient all-reduce must go over RCCL / xGMI (unset RSDET_DIST_BACKEND?)"
                         % (world, torch.cuda.device_count(), dist.get_backend()))


def gather_objects(obj):
    """Every rank's picklable ``obj`` as a list on every rank (the sharded evaluation of Runner.val)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def launch_ranks(nproc, argv, env=None, timeout=None):
    """Start ``nproc`` ranks of ``python argv...`` as CHILD processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
    rendezvous on 127.0.0.1), wait for all of them and return ``(worst exit code, rank 0's stdout)``.

    The mpirun of the reference (README_competition.md:79-80).  Must be called before the calling process has
    initialised the GPU: the children are fresh interpreters, nothing is exec'ed over or forked from a HIP process.
    Ranks 1.. inherit stderr; their stdout is discarded (rank 0 prints the results)."""
    import tempfile
    import time
    base = dict(os.environ if env is None else env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base.update(WORLD_SIZE=str(nproc), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
                LOCAL_WORLD_SIZE=str(nproc))
    procs, rc = [], 0
    with tempfile.TemporaryFile() as out0:
        for r in range(nproc):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable] + list(argv), env=e,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        t0 = time.time()
        try:
            while any(p.poll() is None for p in procs):
                bad = [p.returncode for p in procs if p.poll() not in (None, 0)]
                if bad:
                    rc = bad[0]  # a failed rank leaves the others inside a collective
                    break
                if timeout is not None and time.time() - t0 > timeout:
                    rc = 124
                    break
                time.sleep(0.05)
        finally:
            for p in procs:      # end exactly the processes started here, never by pattern
                if p.poll() is None:
                    p.kill()
                p.wait()
        for p in procs:
            if p.returncode != 0 and rc == 0:
                rc = p.returncode
        out0.seek(0)
        text = out0.read().decode(errors="replace")
    return rc, text
