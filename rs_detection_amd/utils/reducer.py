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
  * a bucket is flushed from the post-accumulate hook of its LAST-to-finish parameter (the first one in registration
    order) if all its gradients are there by then, else at the end of backward: 3 Python hooks per step, not 161;
  * afterwards ``p.grad`` IS the bucket view: the addresses the fused optimizer (csrc/optim.hip) reads never change, so
    its per-step pointer upload disappears as well;
  * the wire dtype is the gradient's own: bf16 parameters (``Runner(bf16_params=True)``) travel as bf16 -- half the
    bytes per xGMI link -- without a compress / decompress pass.
"""
import contextlib

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("flat", "params", "views", "work", "flushed")

    def __init__(self, flat, params, views):
        self.flat, self.params, self.views, self.work, self.flushed = flat, params, views, None, False


class GradReducer:
    def __init__(self, model, bucket_cap_mb=64, process_group=None, broadcast=True):
        assert dist.is_available() and dist.is_initialized(), "GradReducer needs an initialised process group"
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.enabled = True
        params = [p for p in model.parameters() if p.requires_grad]
        if broadcast and self.world > 1:                      # every rank starts from rank 0's weights (as DDP does)
            with torch.no_grad():
                for p in model.parameters():
                    dist.broadcast(p.data, 0, group=process_group)
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
                    self.buckets.append(self._make_bucket(cur, dtype, dev))
                    cur, size = [], 0
                cur.append(p)
                size += nbytes
            if cur:
                self.buckets.append(self._make_bucket(cur, dtype, dev))
        # AVG is RCCL's own reduction; gloo (CPU tests, ranks sharing one GPU) sums and the bucket is scaled afterwards
        self._avg = dist.get_backend(process_group) == "nccl"
        self._hooks = []
        for b in self.buckets:
            sentinel = b.params[-1]                            # first in registration order: its gradient comes last
            self._hooks.append(sentinel.register_post_accumulate_grad_hook(self._make_hook(b)))

    @staticmethod
    def _make_bucket(params, dtype, dev):
        total = sum(p.numel() for p in params)
        flat = torch.zeros((total,), dtype=dtype, device=dev)
        views, off = [], 0
        for p in params:
            # a view with the parameter's own element order (channels_last weights included): the fused optimizer reads
            # gradient and parameter in lock step
            views.append(flat.as_strided(tuple(p.shape), tuple(p.stride()), off))
            off += p.numel()
        return _Bucket(flat, list(params), views)

    def _make_hook(self, bucket):
        def hook(_param):
            if self.enabled and not bucket.flushed and all(p.grad is not None for p in bucket.params):
                self._flush(bucket)
        return hook

    def _flush(self, b):
        src, dst = [], []
        for p, v in zip(b.params, b.views):
            if p.grad is None:
                v.zero_()                                      # an unused parameter contributes zeros to the mean
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
        if dst:
            torch._foreach_copy_(dst, src)
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.group,
                                 async_op=True)
        b.flushed = True

    def reduce(self):
        """After ``backward()``: flush what the hooks could not, wait for the collectives, and make every ``p.grad`` the
        (now averaged) bucket view."""
        if not self.enabled:
            return
        for b in self.buckets:
            if not b.flushed:
                self._flush(b)
        for b in self.buckets:
            b.work.wait()
            if not self._avg and self.world > 1:
                b.flat.mul_(1.0 / self.world)
            for p, v in zip(b.params, b.views):
                p.grad = v
            b.work, b.flushed = None, False

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
