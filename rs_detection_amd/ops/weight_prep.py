"""Prepared weight operands of the bf16 backward, refreshed by ONE launch per optimizer step (csrc/layout.hip:
weight_prep_multi_kernel).

Two families of operands are pure functions of a convolution weight and are needed once per backward:

  * the flipped weights (C, O, 3, 3) of "backward-data through the forward solver" (ops/conv3x3.py), and
  * the transposed weights (C, O) of the 1x1 backward-data GEMM, optionally with the scale of the eval-mode BatchNorm behind
    the convolution folded in (ops/bottleneck.py).

Rounds 3-5 made them where they were used: one small launch (and one allocation) per use, 37 per S2ANet-R50 step.  The
weights only change when an optimizer steps (or a checkpoint loads), so a registry keeps one persistent output per weight
and recomputes ALL of them with one multi-tensor launch the first time any is asked for after a change.

Staleness is detected two ways, both checked on every request: torch's own version counters of the weight and of the folded
BatchNorm's running_var / gamma (bumped by every in-place torch operation: torch optimizers, load_state_dict, copy_; the
BatchNorm tensors are held weakly, an entry whose tensors died is dropped) and a package-wide epoch that our fused optimizers --
which write through raw pointers and so do not touch version counters -- bump in step() (optims/optimizer.py).  Entries hold
the Parameter weakly; a changed data pointer (p.data reassigned, module moved) rebuilds the device table.  The one edit
neither sees is an in-place write through ``p.data`` (its alias has a version counter of its own): call bump_epoch() after it.

The reference has no counterpart: cuDNN derives whatever operand layout it wants inside its backward kernels
(/root/reference/python/jdet/models/backbones/resnet.py:57-93 backward)."""
import struct
import weakref

import torch
from torch.utils.weak import WeakIdKeyDictionary

from .. import _lib

_EPOCH = [0]


def bump_epoch():
    """Called by anything that changes parameter VALUES without going through torch in-place operations."""
    _EPOCH[0] += 1


class Entry:
    __slots__ = ("w", "bn", "flip", "out", "version", "ptrs", "reg", "__weakref__")

    def bn_tensors(self):
        """(running_var, gamma, eps) of the folded BatchNorm (held weakly), or None; a dead reference makes the entry dead."""
        if self.bn is None:
            return None
        var, gamma = self.bn[0](), (self.bn[1]() if self.bn[1] is not None else None)
        if var is None or (self.bn[1] is not None and gamma is None):
            return False
        return var, gamma, self.bn[2]

    def _stamp(self, w):
        bn = self.bn_tensors()
        if not bn:
            return (w._version,)
        var, gamma, _ = bn
        return (w._version, var._version, var.data_ptr(), -1 if gamma is None else gamma._version,
                0 if gamma is None else gamma.data_ptr())

    def tensor(self):
        """The prepared operand, fresh with respect to the current values of the weight AND of the folded BatchNorm's
        running_var / gamma (torch in-place edits of either bump their version counters: a partial load_state_dict, EMA /
        SWA averaging, a manual re-initialisation)."""
        reg = self.reg
        w = self.w()
        if reg.epoch != _EPOCH[0] or w is None or self.ptrs[0] != w.data_ptr() or self.version != self._stamp(w):
            reg.refresh()
        return self.out


class _Registry:
    def __init__(self, device):
        self.device = device
        self.entries = WeakIdKeyDictionary()             # Parameter -> {key: Entry}
        self.table = None
        self.table_key = None
        self.total_tiles = 0
        self.epoch = -1

    def entry(self, w, bn=None, flip=False):
        per = self.entries.get(w)
        if per is None:
            per = self.entries[w] = {}
        key = (None if bn is None else (id(bn[0]), id(bn[1]), float(bn[2])), bool(flip))
        e = per.get(key)
        if e is not None and bn is not None:
            cur = e.bn_tensors()             # (ids can be reused after a tensor died: the entry must hold THESE tensors)
            if not cur or cur[0] is not bn[0] or cur[1] is not bn[1]:
                e = None
        if e is None:
            O, C = w.shape[0], w.shape[1]
            T = w.shape[2] * w.shape[3]
            e = Entry()
            e.w, e.flip, e.reg = weakref.ref(w), flip, self
            e.bn = None if bn is None else (weakref.ref(bn[0]), None if bn[1] is None else weakref.ref(bn[1]), float(bn[2]))
            if T == 1:
                e.out = torch.empty((C, O), dtype=torch.bfloat16, device=w.device)
            else:
                e.out = torch.empty((C, O, w.shape[2], w.shape[3]), dtype=torch.bfloat16, device=w.device,
                                    memory_format=torch.channels_last)
            e.version, e.ptrs = -1, (0,)
            per[key] = e
            self.epoch = -1                              # a new entry: the next request refreshes
        return e

    def _live(self):
        out = []
        for w, per in list(self.entries.items()):
            for key, e in list(per.items()):
                if e.bn_tensors() is False:      # its BatchNorm's tensors are gone (buffer reassigned, module dropped)
                    del per[key]
                    continue
                out.append((w, e))
        return out

    def refresh(self):
        live = self._live()
        key, recs, tile0 = [], [], 0
        for w, e in live:
            var = gamma = None
            eps = 0.0
            if e.bn is not None:
                var, gamma, eps = e.bn_tensors()
            ptrs = (w.data_ptr(), e.out.data_ptr(), 0 if var is None else var.data_ptr(),
                    0 if gamma is None else gamma.data_ptr())
            e.ptrs = ptrs
            O, C = w.shape[0], w.shape[1]
            T = w.shape[2] * w.shape[3]
            tc, to = (C + 31) // 32, (O + 31) // 32
            recs.append(struct.pack("<QQQQiiifiiii", ptrs[0], ptrs[1], ptrs[2], ptrs[3], O, C, T, float(eps), tile0, tc, to, 0))
            key.append(ptrs + (O, C, T, float(eps)))
            tile0 += T * tc * to
        key = tuple(key)
        if key != self.table_key:
            # (first steps and pointer changes only: a host -> device copy of 64 bytes per entry)
            buf = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8) if recs else torch.empty(0, dtype=torch.uint8)
            self.table = buf.to(self.device)
            self.table_key, self.total_tiles = key, tile0
        if live:
            rc = _lib.load().rsdet_weight_prep_multi_bf16(_lib.ptr(self.table), len(live), self.total_tiles,
                                                          _lib.stream_ptr())
            _lib.check(rc, "rsdet_weight_prep_multi_bf16")
        for w, e in live:
            e.version = e._stamp(w)
        self.epoch = _EPOCH[0]


_REGISTRIES = {}


def applies(w):
    """A bf16 CUDA Parameter in channels_last storage (what a Runner(bf16_params=True) model holds): (O, T, C) in memory."""
    return (isinstance(w, torch.nn.Parameter) and w.is_cuda and w.dtype == torch.bfloat16 and w.dim() == 4
            and w.is_contiguous(memory_format=torch.channels_last))


def entry(w, bn=None, flip=False):
    """The registry entry of weight ``w`` (a Parameter that applies()): ``bn = (running_var, gamma, eps)`` folds that
    BatchNorm's scale into a 1x1 weight's transpose; ``flip`` marks the 3x3 flipped-weights operand (the tap reversal is what
    the kernel does for every entry; for T = 1 it is the identity)."""
    reg = _REGISTRIES.get(w.device)
    if reg is None:
        reg = _REGISTRIES[w.device] = _Registry(w.device)
    return reg.entry(w, bn, flip)
