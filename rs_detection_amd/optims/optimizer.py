"""SGD / AdamW with JDet's config surface (/root/reference/python/jdet/optims/optimizer.py:24-43):
``grad_clip=dict(max_norm=35, norm_type=2)`` clips the global grad norm before the update.
Jittor's ``optimizer.step(loss)`` = backward + (MPI all-reduce) + clip + update; here
backward/all-reduce belong to torch autograd + DDP (RCCL), ``step()`` clips and updates.
Fused multi-tensor updates (one launch per parameter group, not per parameter)."""
import torch

from rs_detection_amd.utils.registry import OPTIMS


class _Mixin:
    def parameters_dict(self):
        return self.state_dict()

    def cur_lr(self):
        return self.param_groups[0]["lr"]

    def _clip(self):
        if getattr(self, "grad_clip", None):
            params = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
            torch.nn.utils.clip_grad_norm_(params, float(self.grad_clip.get("max_norm", 35)),
                                           float(self.grad_clip.get("norm_type", 2)), foreach=True)


@OPTIMS.register_module()
class SGD(torch.optim.SGD, _Mixin):
    def __init__(self, params, lr, momentum=0, weight_decay=0, dampening=0, nesterov=False, grad_clip=None):
        params = [p for p in params if p.requires_grad] if not isinstance(params, (list, tuple)) or \
            (params and not isinstance(params[0], dict)) else params
        super().__init__(params, lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=dampening,
                         nesterov=nesterov, foreach=True)
        self.grad_clip = grad_clip
        self.lr = lr

    def step(self, closure=None):
        self._clip()
        return super().step(closure)


@OPTIMS.register_module()
class AdamW(torch.optim.AdamW, _Mixin):
    def __init__(self, params, lr, eps=1e-8, betas=(0.9, 0.999), weight_decay=0, grad_clip=None):
        params = [p for p in params if p.requires_grad] if not isinstance(params, (list, tuple)) or \
            (params and not isinstance(params[0], dict)) else params
        super().__init__(params, lr=lr, eps=eps, betas=tuple(betas), weight_decay=weight_decay, foreach=True)
        self.grad_clip = grad_clip
        self.lr = lr

    def step(self, closure=None):
        self._clip()
        return super().step(closure)


@OPTIMS.register_module()
class FusedSGD(torch.optim.Optimizer, _Mixin):
    """SGD + global grad-norm clip as TWO launches over all parameters (csrc/optim.hip), with fp32 MASTER copies for
    parameters the model holds in bf16 (``Runner(bf16_params=True)``): the kernel reads bf16 / fp32 gradients, updates
    the fp32 master and momentum and writes the model's copy in its own dtype -- no autocast weight cast, no gradient
    widening, no foreach passes.  Same arithmetic as ``SGD`` above (torch.optim.SGD with dampening 0, no nesterov;
    clip_grad_norm_ with norm_type 2), which the reference configures in optims/optimizer.py:24-43.

    One hyper-parameter set for all parameters (the reference's configs have a single group).  ``state_dict`` carries
    ``momentum_buffer`` and, for bf16 parameters, ``master``."""

    def __init__(self, params, lr, momentum=0, weight_decay=0, dampening=0, nesterov=False, grad_clip=None):
        assert dampening == 0 and not nesterov, "FusedSGD: dampening / nesterov are not on the reference's path"
        params = [p for p in params if p.requires_grad] if not isinstance(params, (list, tuple)) or \
            (params and not isinstance(params[0], dict)) else params
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        assert len(self.param_groups) == 1, "FusedSGD: one parameter group"
        self.grad_clip = grad_clip
        self.lr = lr
        self._params_key = None

    def load_state_dict(self, state_dict):
        """torch's ``load_state_dict`` casts floating-point state to the PARAMETER's dtype: for a bf16 parameter that
        would round the fp32 master and momentum (and hand the kernel 2-byte buffers).  The state of this optimizer is
        fp32 whatever the parameter holds: put the saved tensors back unrounded."""
        super().load_state_dict(state_dict)
        saved_ids = [i for g in state_dict["param_groups"] for i in g["params"]]
        params = [p for g in self.param_groups for p in g["params"]]
        for sid, p in zip(saved_ids, params):
            st = state_dict["state"].get(sid)
            if st is None:
                continue
            for k, v in st.items():
                if k != "step" and isinstance(v, torch.Tensor) and v.is_floating_point():
                    self.state[p][k] = v.detach().to(device=p.device, dtype=torch.float32).clone()
        self._params_key = None          # the device table points at the old state tensors

    _STATE_KEYS = ("momentum_buffer",)       # fp32 companions of a parameter, in record order (mom, mom2)

    def _ensure_state(self, p):
        st = self.state[p]
        for k in self._STATE_KEYS:
            if k not in st:
                st[k] = torch.zeros(p.shape, dtype=torch.float32, device=p.device)
        if p.dtype == torch.bfloat16 and "master" not in st:
            st["master"] = p.detach().float().clone()
        return st

    def master_state_dict(self, model):
        """name -> fp32 tensor for every parameter of ``model`` (the master where the model holds bf16): what a
        checkpoint stores, so that a saved run continues from the unrounded weights."""
        out = {}
        frozen = getattr(self, "frozen_masters", None) or {}
        for name, p in model.named_parameters():
            st = self.state.get(p, {})
            # a frozen weight the Runner cast to bf16 has no optimizer state: its fp32 original is kept in
            # ``frozen_masters`` (set by the Runner), so a save / load cycle does not round the pretrained values
            out[name] = st["master"] if "master" in st else frozen.get(name, p.detach())
        return out

    def set_masters(self, model, params):
        """After a checkpoint went into a model with bf16 weights: the fp32 values of ``params`` (name -> ndarray /
        tensor) become the masters, so that the run continues from the unrounded weights."""
        import numpy as np
        named = dict(model.named_parameters())
        with torch.no_grad():
            frozen = getattr(self, "frozen_masters", None)
            for k, v in params.items():
                p = named.get(k)
                if p is None or p.dtype != torch.bfloat16:
                    continue
                t = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else torch.as_tensor(v)
                if tuple(t.shape) != tuple(p.shape):
                    continue
                if not p.requires_grad:
                    if frozen is not None and k in frozen:
                        frozen[k].copy_(t.to(device=p.device, dtype=torch.float32))
                    continue
                st = self._ensure_state(p)
                st["master"].copy_(t.to(device=p.device, dtype=torch.float32))

    @staticmethod
    def _same_order(a, b):
        """Same element order in memory: strides equal on every dimension of size > 1."""
        return a.shape == b.shape and all(sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()) if n > 1)

    def _build(self, params, lib):
        """Static part of the device table (once per parameter set): records without the gradient pointers + chunks."""
        import numpy as np
        dev = params[0].device
        chunk = lib.rsdet_mt_chunk_elems()
        rec = np.zeros((len(params), 8), np.int64)              # 64-byte records of csrc/optim.hip (MtTensor)
        chunks = []
        for i, p in enumerate(params):
            assert p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last), "dense parameters only"
            st = self._ensure_state(p)
            for k in self._STATE_KEYS + ("master",):            # companions in the parameter's memory order
                t = st.get(k)
                if t is not None and not self._same_order(t, p):
                    st[k] = torch.empty_strided(p.shape, p.stride(), dtype=torch.float32, device=dev).copy_(t)
            master = st.get("master")
            rec[i, 1:5] = (p.data_ptr(), master.data_ptr() if master is not None else 0,
                           st[self._STATE_KEYS[0]].data_ptr(), p.numel())
            if len(self._STATE_KEYS) > 1:
                rec[i, 6] = st[self._STATE_KEYS[1]].data_ptr()
            rec[i, 5] = 2 if p.dtype == torch.bfloat16 else 0    # low 32 bits = flags (little endian); grad bit per step
            chunks += [(i, c) for c in range((p.numel() + chunk - 1) // chunk)]
        self._rec = rec
        self._gflag = np.array([1 if p.dtype == torch.bfloat16 else 0 for p in params], np.int64)
        n_rec32 = rec.size * 2
        host = np.concatenate([rec.reshape(-1).view(np.int32), np.asarray(chunks, np.int32).reshape(-1)])
        # The records go up once; per step only the gradient pointers change.  Staging for that upload: a RING of pinned
        # copies of the record block, each guarded by an event -- an async copy out of ONE pinned buffer that the next
        # step rewrites races with a GPU that is a step or more behind the host (the fp32 step is: found as a
        # non-finite loss in the bench), and a pageable source makes the copy synchronise the stream (24.9 vs 22.4 ms).
        self._table = torch.empty((host.size,), dtype=torch.int32, device=dev)
        self._table.copy_(torch.from_numpy(host.copy()))
        self._ring = [torch.empty((n_rec32,), dtype=torch.int32, pin_memory=True) for _ in range(8)]
        for r in self._ring:
            r.numpy()[...] = host[:n_rec32]
        self._ring_ev, self._ring_at = [None] * len(self._ring), 0
        self._n_rec32, self._n_chunks = n_rec32, len(chunks)
        self._state_buf = torch.zeros((lib.rsdet_mt_sgd_state_bytes(self._n_chunks),), dtype=torch.uint8, device=dev)
        self._params_key = tuple(id(p) for p in params)
        self._grad_ptrs = None

    def zero_grad(self, set_to_none=True):
        """torch.optim.Optimizer.zero_grad's semantics without its per-parameter bookkeeping (1 ms per Oriented R-CNN step):
        gradients dropped (set_to_none) or zeroed in place."""
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for grp in self.param_groups:
            for p in grp["params"]:
                p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        from rs_detection_amd import _lib
        lib = _lib.load()
        g = self.param_groups[0]
        # ONE pass over the parameters (700 of them for VAN-B3: `p.grad` is a property call, and three passes with a generic
        # layout check cost 2.9 ms of host time per Oriented R-CNN step): the parameters that have a gradient, and the
        # gradient pointers -- autograd hands out fresh gradient tensors after zero_grad(set_to_none=True); DDP's /
        # the reducer's bucket views stay put and skip the upload below
        params, ptrs = [], []
        for p in g["params"]:
            gr = p.grad
            if gr is None:
                continue
            # (equal strides = same element order; only a gradient whose strides differ takes the dimension-wise check)
            if gr.dtype != p.dtype or (gr.stride() != p.stride() and not self._same_order(gr, p)):
                gr = p.grad = torch.empty_strided(p.shape, p.stride(), dtype=p.dtype, device=p.device).copy_(gr)
            params.append(p)
            ptrs.append(gr.data_ptr())
        if not params:
            return None
        if getattr(self, "_params_key", None) != tuple(map(id, params)):
            self._build(params, lib)
        if ptrs != self._grad_ptrs:
            import numpy as np
            k = self._ring_at
            self._ring_at = (k + 1) % len(self._ring)
            if self._ring_ev[k] is not None:
                self._ring_ev[k].synchronize()        # the copy issued 8 steps ago out of this buffer: long done
            rec = self._ring[k].numpy().view(np.int64).reshape(len(params), 8)
            rec[:, 0] = ptrs
            rec[:, 5] = self._rec[:, 5] | self._gflag
            self._table[:self._n_rec32].copy_(self._ring[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._ring_ev[k] = ev
            self._grad_ptrs = ptrs
        clip = float(self.grad_clip.get("max_norm", 35)) if getattr(self, "grad_clip", None) else 0.0
        if clip > 0:
            assert float(self.grad_clip.get("norm_type", 2)) == 2.0, "FusedSGD clips the L2 norm"
        rc = self._launch(lib, _lib, g, clip)
        if rc != _lib.RSDET_OK:
            self._state_buf.zero_()
        _lib.check(rc, self._ENTRY)
        # the kernel wrote the parameters through raw pointers (no torch version counter moved): operands derived from
        # the weights (ops/weight_prep.py) are stale from here on
        from rs_detection_amd.ops.weight_prep import bump_epoch
        bump_epoch()
        return None

    _ENTRY = "rsdet_mt_sgd_step"

    def _launch(self, lib, _lib, g, clip):
        tab = self._table
        return lib.rsdet_mt_sgd_step(_lib.ptr(tab), ctypes_ptr_offset(tab, self._n_rec32 * 4), self._n_chunks, clip,
                                     float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]), None,
                                     _lib.ptr(self._state_buf), self._state_buf.numel(), _lib.stream_ptr())


@OPTIMS.register_module()
class FusedAdamW(FusedSGD):
    """AdamW + global grad-norm clip as two launches over all parameters (csrc/optim.hip: mt_adamw_kernel) -- the
    optimizer of configs/orcnn/orcnn_van3_7_anchor.py.  torch.optim.AdamW's arithmetic (amsgrad off): decoupled weight
    decay, exp_avg / exp_avg_sq in fp32, bias corrections from the step count (kept in the parameter group: one count
    for all parameters, as they are all updated every step).  Replaces ~130 foreach launches per step (2.9 ms on
    VAN-B3) with 1-2; bf16 parameters get fp32 masters exactly as in FusedSGD."""

    _STATE_KEYS = ("exp_avg", "exp_avg_sq")
    _ENTRY = "rsdet_mt_adamw_step"

    def __init__(self, params, lr, eps=1e-8, betas=(0.9, 0.999), weight_decay=0, grad_clip=None):
        params = [p for p in params if p.requires_grad] if not isinstance(params, (list, tuple)) or \
            (params and not isinstance(params[0], dict)) else params
        torch.optim.Optimizer.__init__(self, params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay,
                                                          step=0))
        assert len(self.param_groups) == 1, "FusedAdamW: one parameter group"
        self.grad_clip = grad_clip
        self.lr = lr
        self._params_key = None

    # -- state interchange with torch.optim.AdamW (a checkpoint written by one resumes under the other: the Runner
    #    picks the fused form on a GPU and torch's on the CPU).  torch keeps the step count PER PARAMETER
    #    (``state[p]['step']``, a float32 scalar), this class once per group: ``state_dict`` writes the group's count
    #    into every parameter's state, ``load_state_dict`` reads it back from there when the saved group has none.
    def state_dict(self):
        g = self.param_groups[0]
        for p in g["params"]:
            if p.requires_grad:
                self._ensure_state(p)["step"] = torch.tensor(float(g.get("step", 0)), dtype=torch.float32)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        g = self.param_groups[0]
        if not any("step" in sg for sg in state_dict["param_groups"]):
            counts = [int(st["step"]) for st in state_dict["state"].values() if "step" in st]
            g["step"] = max(counts) if counts else 0
        g["step"] = int(g.get("step", 0))

    def _launch(self, lib, _lib, g, clip):
        g["step"] = int(g.get("step", 0)) + 1
        tab = self._table
        return lib.rsdet_mt_adamw_step(_lib.ptr(tab), ctypes_ptr_offset(tab, self._n_rec32 * 4), self._n_chunks, clip,
                                       float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                       float(g["weight_decay"]), g["step"], None, _lib.ptr(self._state_buf),
                                       self._state_buf.numel(), _lib.stream_ptr())


def ctypes_ptr_offset(t, nbytes):
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + nbytes)
