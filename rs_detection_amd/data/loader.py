"""Loader workers and device prefetch for the iterable datasets (f2; /root/reference/python/jdet/data/custom.py:34-35
hands ``num_workers`` to Jittor's multi-process ``Dataset``).

* ``num_workers > 0``: the per-image work (PIL decode + transforms) runs in WORKER PROCESSES.  They come from a
  ``forkserver`` context -- children of a clean server process, never forks of a parent that may already hold a HIP
  context (forking such a process is what takes GPU boxes down) -- and they hide the GPU from themselves.  A sliding
  window keeps ``prefetch`` batches in flight, results come back IN ORDER.
* Determinism: every sample is produced under ``random.seed / np.random.seed(sample_seed(seed, epoch, index))`` -- in
  the worker processes and in the in-process path alike -- so the batches of ``num_workers = N`` equal the batches of
  ``num_workers = 0`` bit for bit, whatever the scheduling (tests/test_loader_cpu.py).
* ``prefetch_to_device``: a feeder thread collates into PINNED host buffers and issues the host-to-device copies on a
  side stream; the training stream only waits for the copy's event, so decode, collate and PCIe overlap the step."""
import multiprocessing as mp
import os
import collections
import queue
import random
import threading

import numpy as np

_WORKER_DS = None
_WORKER_SHM = None      # (SharedMemory, n_slots, slot_bytes) attached in the worker


def sample_seed(seed, epoch, idx):
    return (int(seed) * 1000003 + int(epoch) * 7919 + int(idx) * 104729 + 12345) % (2 ** 32)


def load_sample(ds, epoch, idx):
    """``ds[idx]`` under the sample's own random state (the transforms draw from ``random`` / ``np.random``)."""
    s = sample_seed(getattr(ds, "seed", 0), epoch, idx)
    random.seed(s)
    np.random.seed(s)
    return ds[int(idx)]


def _init_worker(ds, shm_name, n_slots, slot_bytes):
    global _WORKER_DS, _WORKER_SHM
    for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        os.environ[k] = ""            # a worker never touches the GPU
    os.environ["OMP_NUM_THREADS"] = "1"
    _WORKER_DS = ds
    if shm_name:
        from multiprocessing import shared_memory
        # (attaching registers the name with the resource tracker the workers share with the parent -- a set, so the
        #  parent's unlink() clears it once for all)
        _WORKER_SHM = (shared_memory.SharedMemory(name=shm_name), n_slots, slot_bytes)


def _work(task):
    """-> (image, anno), the image either as an ndarray (pickled through the pipe: 12.6 MB for a 1024^2 float tile, 28 ms
    to pickle + 7 ms to unpickle IN THE PARENT) or as ("shm", slot, shape, dtype) when it fits the task's slot of the
    shared-memory ring -- then only the annotation dict travels through the pipe."""
    epoch, idx, slot = task
    image, anno = load_sample(_WORKER_DS, epoch, idx)
    if _WORKER_SHM is not None and slot >= 0 and isinstance(image, np.ndarray):
        shm, n_slots, slot_bytes = _WORKER_SHM
        if image.nbytes <= slot_bytes:
            view = np.ndarray(image.shape, image.dtype, buffer=shm.buf, offset=slot * slot_bytes)
            view[...] = image
            return ("shm", slot, image.shape, image.dtype.str), anno
    return image, anno


class WorkerPool:
    """Persistent pool of ``n`` loader processes holding a pickled copy of the dataset, plus a ring of shared-memory
    slots the workers write their images into.  Slot of task t = t mod n_slots with n_slots = window + 2 batches: a
    slot is handed out again only after ``n_slots`` further tasks were submitted, and tasks are submitted one per
    consumed result -- by then the consumer has collated the batch that image belonged to."""

    def __init__(self, ds, n, window, slot_bytes=None):
        try:
            ctx = mp.get_context("forkserver")
            ctx.set_forkserver_preload(["numpy", "PIL.Image", "rs_detection_amd.data"])
        except ValueError:            # platform without forkserver
            ctx = mp.get_context("spawn")
        self.n, self.window = int(n), int(window)
        self.n_slots = self.window + 2 * max(int(getattr(ds, "batch_size", 1)), 1)
        self.shm, self.slot_bytes = None, 0
        if slot_bytes is None:        # size of one sample, with headroom for multi-scale pipelines
            try:
                probe = load_sample(ds, 0, 0)[0]
                slot_bytes = int(probe.nbytes * 1.3) if isinstance(probe, np.ndarray) else 0
            except Exception:
                slot_bytes = 0
        if slot_bytes > 0:
            from multiprocessing import shared_memory
            self.slot_bytes = (int(slot_bytes) + 4095) & ~4095
            self.shm = shared_memory.SharedMemory(create=True, size=self.slot_bytes * self.n_slots)
        self.pool = ctx.Pool(self.n, initializer=_init_worker,
                             initargs=(ds, self.shm.name if self.shm else None, self.n_slots, self.slot_bytes))
        self._task = 0

    def _submit(self, epoch, idx):
        slot = self._task % self.n_slots if self.shm is not None else -1
        self._task += 1
        return self.pool.apply_async(_work, ((epoch, idx, slot),))

    def _resolve(self, res):
        image, anno = res
        if isinstance(image, tuple) and image and image[0] == "shm":
            _, slot, shape, dt = image
            image = np.ndarray(shape, np.dtype(dt), buffer=self.shm.buf, offset=slot * self.slot_bytes)
        return image, anno

    def imap_window(self, tasks, window=None):
        """Ordered results over ``tasks`` = (epoch, idx) pairs with at most ``window`` in flight.  Images may be VIEWS
        into the shared ring: valid until ``2 * batch_size`` further results have been taken (collate before that)."""
        window = min(window or self.window, self.window)
        pending = []
        it = iter(tasks)
        for t in it:
            pending.append(self._submit(*t))
            if len(pending) >= window:
                break
        while pending:
            res = self._resolve(pending.pop(0).get())
            nxt = next(it, None)
            if nxt is not None:
                pending.append(self._submit(*nxt))
            yield res

    def close(self):
        if self.pool is not None:
            self.pool.terminate()
            self.pool.join()
            self.pool = None
        if self.shm is not None:
            try:
                self.shm.close()
                self.shm.unlink()     # /dev/shm is memory: never leave the ring behind
            except Exception:
                pass
            self.shm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_pinned_ok = None


def reusable_batch_buffer(owner, shape, dtype=np.float32, ring=6):
    """One of ``ring`` long-lived (n, 3, H, W) arrays of this shape, handed out round-robin: collating into a fresh
    ``np.zeros`` page-faults 50 MB per batch (measured 0.5-0.9 s in a container, tens of ms on a quiet host) -- and,
    where a GPU is present, PINNED, so that the host-to-device copy is asynchronous without another staging copy.
    A batch stays valid until ``ring - 1`` further batches have been produced."""
    global _pinned_ok
    ringd = owner.__dict__.setdefault("_batch_ring", {})
    key = (tuple(shape), np.dtype(dtype).str)
    ent = ringd.get(key)
    if ent is None:
        bufs = []
        for _ in range(ring):
            arr = None
            if _pinned_ok is not False:
                try:
                    import torch
                    if torch.cuda.is_available():
                        t = torch.empty(tuple(shape), dtype=torch.from_numpy(np.empty(0, dtype)).dtype, pin_memory=True)
                        arr = t.numpy()
                        owner.__dict__.setdefault("_batch_ring_keep", []).append(t)
                        _pinned_ok = True
                    else:
                        _pinned_ok = False
                except Exception:
                    _pinned_ok = False
            bufs.append(arr if arr is not None else np.empty(tuple(shape), dtype))
        ent = ringd[key] = [bufs, 0]
        if len(ringd) > 8:            # multi-scale pipelines: keep the most recent shapes only
            ringd.pop(next(iter(ringd)))
    bufs, k = ent
    ent[1] = (k + 1) % len(bufs)
    return bufs[k]


def iterate_samples(ds, indices, prefetch_batches=4):
    """The samples of ``indices`` in order: through the dataset's worker pool when ``ds.num_workers > 0`` (created on
    first use, kept for later epochs), in-process otherwise.  Same values either way."""
    n = int(getattr(ds, "num_workers", 0) or 0)
    epoch = int(getattr(ds, "epoch", 0))
    if n <= 0:
        for i in indices:
            yield load_sample(ds, epoch, int(i))
        return
    window = max(int(getattr(ds, "batch_size", 1)) * prefetch_batches, 2 * n)
    pool = ds.__dict__.get("_worker_pool")
    if pool is None or pool.n != n or pool.pool is None or pool.window < window:
        if pool is not None:
            pool.close()
        pool = ds.__dict__["_worker_pool"] = WorkerPool(ds, n, window)
    yield from pool.imap_window(((epoch, int(i)) for i in indices), window)


def state_without_pool(obj):
    """``__getstate__`` helper: the pool itself never travels to the workers."""
    d = dict(obj.__dict__)
    for k in ("_worker_pool", "_batch_ring", "_batch_ring_keep"):
        d.pop(k, None)
    return d


def prefetch_to_device(dataset, device, depth=2):
    """``for images, targets in prefetch_to_device(ds, device)``: device tensors, produced ``depth`` batches ahead by a
    feeder thread (dataset iteration + collate -> pinned buffer -> ``non_blocking`` copy on a side stream)."""
    import torch
    from . import batch_to_device
    dev = torch.device(device)
    if dev.type != "cuda":
        for images, targets in dataset:
            yield batch_to_device(images, targets, dev)
        return
    side = torch.cuda.Stream(device=dev)
    q = queue.Queue(maxsize=depth)
    had = getattr(dataset, "reuse_batch_buffers", None)
    try:
        dataset.reuse_batch_buffers = True    # collate straight into pinned, long-lived buffers (no page faults, no staging copy)
    except Exception:
        pass
    stop = threading.Event()
    pinned = {}

    def feeder():
        try:
            torch.cuda.set_device(dev)
            slot = 0
            copied = collections.deque()   # events of the batches whose pinned source may still be read by the side stream
            it = iter(dataset)
            while True:
                # A pinned source buffer (the dataset's collate ring of 6, the staging ring of depth + 2 below) is
                # rewritten a few batches later: the copy that read it three batches ago must be over first.
                while len(copied) >= 3:
                    copied.popleft().synchronize()
                try:
                    images, targets = next(it)
                except StopIteration:
                    break
                if stop.is_set():
                    return
                arr = np.ascontiguousarray(images)
                buf = torch.from_numpy(arr)
                if not buf.is_pinned():       # datasets that collate into reusable_batch_buffer() hand over pinned memory
                    key = (slot % (depth + 2), arr.shape, arr.dtype.str)
                    stage = pinned.get(key)
                    if stage is None:
                        stage = pinned[key] = torch.empty(arr.shape, dtype=buf.dtype, pin_memory=True)
                    stage.numpy()[...] = arr
                    buf = stage
                with torch.cuda.stream(side):
                    img_d = buf.to(dev, non_blocking=True)
                    tg_d = []
                    for t in targets:
                        t = dict(t)
                        for k in ("rboxes", "hboxes", "polys", "labels", "rboxes_ignore"):
                            if isinstance(t.get(k), np.ndarray):
                                t[k] = torch.from_numpy(np.ascontiguousarray(t[k])).to(dev, non_blocking=True)
                        tg_d.append(t)
                    ev = torch.cuda.Event()
                    ev.record(side)
                copied.append(ev)
                q.put((img_d, tg_d, ev))
                slot += 1
            q.put(None)
        except BaseException as e:     # surfaces in the consumer
            q.put(e)

    th = threading.Thread(target=feeder, daemon=True)
    th.start()
    try:
        while True:
            item = q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            img_d, tg_d, ev = item
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ev)
            # Every tensor allocated on the side stream is used on the consumer's stream: without record_stream the
            # caching allocator hands a dropped block back to the side-stream pool at once, and the feeder -- two or
            # three batches ahead of a GPU that itself runs behind the host -- would overwrite targets that queued
            # kernels have not read yet.
            img_d.record_stream(cur)
            for t in tg_d:
                for v in t.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(cur)
            yield img_d, tg_d
    finally:
        stop.set()
        try:
            if had is None:
                del dataset.reuse_batch_buffers
            else:
                dataset.reuse_batch_buffers = had
        except Exception:
            pass
        while th.is_alive():            # unblock a feeder waiting on a full queue
            try:
                q.get_nowait()
            except queue.Empty:
                pass
            th.join(timeout=0.05)
