"""Gradient mean over the data-parallel ranks, bucketed, overlapped with backward -- the collective of the reference's
optimizer step (/root/reference/python/jdet/optims/optimizer.py:30-31: every gradient all-reduced with op "mean" under
MPI; utils/general.py:30-48 for the logged scalars) -- sized for a step that is paced by its HOST.

Why not ``torch.nn.parallel.DistributedDataParallel`` (kept as ``utils/dist.wrap_ddp`` and compared with in the tests):
with ``zero_grad(set_to_none=True)`` autograd hands every parameter a fresh gradient tensor each step and DDP's reducer
copies each one into its bucket view -- one copy launch per parameter, 161 per S2ANet step, plus a C++ hook per
parameter and a Python comm hook per bucket.  Measured on the bf16 S2ANet step under an ``nccl`` group of one rank
(bench.py, ``ddp`` object): +1.6 ms of host time on a 15.5 ms host-paced step.  Here:

  * the parameters are packed, in REVERSE registration order (roughly the order backward produces gradients in), into
    flat buckets of <= ``bucket_cap_mb`` per dtype; a bucket's gradients go into it with ONE multi-tensor copy
    (``torch._foreach_copy_``) followed by ONE asynchronous all-reduce (RCCL over xGMI: few, large messages);
  * afterwards ``p.grad`` IS the bucket view: the addresses the fused optimizer (csrc/optim.hip) reads never change, so
    its per-step pointer upload disappears as well;
  * the wire dtype is the gradient's own: bf16 parameters (``Runner(bf16_params=True)``) travel as bf16 -- half the
    bytes per xGMI link -- without a compress / decompress pass.

The collective schedule is the SAME on every rank by construction (round 5's was data-dependent):

  * buckets are launched in INDEX ORDER, each exactly once per step: a hook may launch bucket i only when buckets 0 .. i-1
    are launched; ``reduce()`` launches whatever is left, in index order.  No rank can issue bucket j before bucket i < j.
  * a hook launches a bucket early only in an ARMED step -- ``begin_step()`` (the Runner calls it between ``zero_grad`` and
    ``backward``) found every bucketed gradient ``None``, so "gradient present" means "produced by THIS backward".  Any
    other start state (gradients kept by ``zero_grad(set_to_none=False)``, local sums left by ``no_sync()``, a caller that
    never calls ``begin_step``) sends every bucket at the end of backward instead: correct, just not overlapped.
  * which parameters receive a gradient is STATIC (DDP's ``static_graph`` contract), and checked: the first ``reduce()``
    records the set, all-reduces it (one tiny MAX collective, the only host synchronisation this class ever makes) and
    raises ON EVERY RANK if the ranks disagree; every later step checks the local set against the record and raises on
    deviation.  A parameter no rank ever uses is left out: its ``p.grad`` stays ``None`` (no decay / momentum on it).
"""
import contextlib

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("flat", "params", "views", "names", "work", "flushed", "used")

    def __init__(self, flat, params, views, names):
        self.flat, self.params, self.views, self.names = flat, params, views, names
        self.work, self.flushed = None, False
        self.used = None                     # per parameter: does it receive a gradient (fixed by the first reduce())


class GradReducer:
    def __init__(self, model, bucket_cap_mb=64, process_group=None, broadcast=True):
        assert dist.is_available() and dist.is_initialized(), "GradReducer needs an initialised process group"
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.enabled = True
        names = {id(p): n for n, p in model.named_parameters()}
        params = [p for p in model.parameters() if p.requires_grad]
        if broadcast and self.world > 1:
            # every rank starts from rank 0's weights AND buffers (BatchNorm running statistics), as DDP does
            with torch.no_grad():
                for t in model.state_dict().values():
                    if isinstance(t, torch.Tensor):
                        dist.broadcast(t.data, 0, group=process_group)
            # (a write through .data moves no version counter: operands derived from the weights are stale now)
            from rs_detection_amd.ops.weight_prep import bump_epoch
            bump_epoch()
        cap = int(bucket_cap_mb * (1 << 20))
        self.buckets, by_dtype = [], {}
        for p in reversed(params):
            by_dtype.setdefault((p.dtype, p.device), []).append(p)
        for (dtype, dev), ps in by_dtype.items():
            cur, size = [], 0
            for p in ps:
                nbytes = p.numel() * p.element_size()
                if cur and size + nbytes > cap:
                    self.buckets.append(self._make_bucket(cur, dtype, dev, names))
                    cur, size = [], 0
                cur.append(p)
                size += nbytes
            if cur:
                self.buckets.append(self._make_bucket(cur, dtype, dev, names))
        # AVG is RCCL's own reduction; gloo (CPU tests, ranks sharing one GPU) sums and the bucket is scaled afterwards
        self._avg = dist.get_backend(process_group) == "nccl"
        self._hooks = []
        self._next = 0                       # the next bucket (index order) to launch this step
        self._armed = False                  # begin_step() found every gradient None: hooks may launch early
        self._static = False                 # the used-parameter record exists (set by the first reduce())

    @staticmethod
    def _make_bucket(params, dtype, dev, names):
        total = sum(p.numel() for p in params)
        flat = torch.zeros((total,), dtype=dtype, device=dev)
        views, off = [], 0
        for p in params:
            # a view with the parameter's own element order (channels_last weights included): the fused optimizer reads
            # gradient and parameter in lock step
            views.append(flat.as_strided(tuple(p.shape), tuple(p.stride()), off))
            off += p.numel()
        return _Bucket(flat, list(params), views, [names.get(id(p), "<unnamed>") for p in params])

    # ---- the step protocol ---------------------------------------------------------------------------------------------
    def begin_step(self):
        """Between ``zero_grad`` and ``backward``.  Arms the early (overlapped) launches when every bucketed gradient is
        ``None`` -- the only start state in which a hook can tell this step's gradients from leftovers."""
        self._armed = False
        if not (self.enabled and self._static):
            return
        for b in self.buckets:
            for p in b.params:
                if p.grad is not None:
                    return
        self._armed = True

    def _ready(self, b):
        for p, u in zip(b.params, b.used):
            if u and p.grad is None:
                return False
        return True

    def _on_sentinel(self, _param):
        if not (self.enabled and self._armed):
            return
        # index order, each bucket once: launch the run of ready buckets that starts at the first unlaunched one
        while self._next < len(self.buckets) and self._ready(self.buckets[self._next]):
            self._flush(self.buckets[self._next])

    def _flush(self, b):
        assert not b.flushed and b is self.buckets[self._next], "buckets are launched in index order, once per step"
        src, dst = [], []
        for p, v, u, name in zip(b.params, b.views, b.used, b.names):
            g = p.grad
            if (g is not None) != u:
                raise RuntimeError(
                    "GradReducer: parameter %r %s a gradient in this step but %s in the first one -- which parameters are "
                    "used must not change between steps (the collective schedule and the bucket contents are static)"
                    % (name, "has" if g is not None else "has NOT", "did not" if g is not None else "did"))
            if u and g.data_ptr() != v.data_ptr():
                src.append(g)
                dst.append(v)
        if dst:
            torch._foreach_copy_(dst, src)
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.group,
                                 async_op=True)
        b.flushed = True
        self._next += 1

    def _record_usage(self):
        """First reduce(): fix which parameters receive gradients and verify every rank agrees (raises on all ranks)."""
        bits = [[p.grad is not None for p in b.params] for b in self.buckets]
        flat = [int(x) for row in bits for x in row]
        if self.world > 1 and flat:
            dev = self.buckets[0].flat.device
            t = torch.tensor([flat, [1 - x for x in flat]], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            both = (t[0] * t[1]).cpu().tolist()               # 1 where some rank used the parameter and some did not
            if any(both):
                names = [n for b in self.buckets for n in b.names]
                bad = [n for n, x in zip(names, both) if x]
                raise RuntimeError("GradReducer: the ranks disagree on which parameters receive a gradient (%d of them, "
                                   "e.g. %s): the collective schedule would differ between ranks" % (len(bad), bad[:4]))
        for b, row in zip(self.buckets, bits):
            b.used = row
            # the sentinel: the LAST used parameter to get its gradient = the first one in registration order
            last = [p for p, u in zip(b.params, row) if u]
            if last:
                self._hooks.append(last[-1].register_post_accumulate_grad_hook(self._on_sentinel))
        self._static = True

    def reduce(self):
        """After ``backward()``: launch what the hooks did not (index order), wait for the collectives, and make every
        used ``p.grad`` the (now averaged) bucket view."""
        if not self.enabled:
            return
        if not self._static:
            self._record_usage()
        while self._next < len(self.buckets):
            self._flush(self.buckets[self._next])
        for b in self.buckets:
            b.work.wait()
            if not self._avg and self.world > 1:
                b.flat.mul_(1.0 / self.world)
            b.work, b.flushed = None, False
        self._next, self._armed = 0, False
        for b in self.buckets:
            for p, v, u, name in zip(b.params, b.views, b.used, b.names):
                if u:
                    p.grad = v
                elif p.grad is not None:     # (its bucket may have left before this gradient appeared: checked here)
                    raise RuntimeError("GradReducer: parameter %r has a gradient in this step but did not in the first one -- "
                                       "which parameters are used must not change between steps" % name)

    @contextlib.contextmanager
    def no_sync(self):
        """Gradients stay local inside the block (DDP's ``no_sync``)."""
        prev, self.enabled = self.enabled, False
        try:
            yield
        finally:
            self.enabled = prev

    @property
    def wire_dtypes(self):
        return sorted({str(b.flat.dtype) for b in self.buckets})

    def owns(self, grad):
        """Is ``grad`` a view into one of the buckets?"""
        return any(b.flat.data_ptr() <= grad.data_ptr() < b.flat.data_ptr() + b.flat.numel() * b.flat.element_size()
                   for b in self.buckets)
