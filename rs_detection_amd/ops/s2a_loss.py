"""Focal + smooth-L1 of one S2ANet module over all pyramid levels, one HIP pass each way (csrc/losses.hip).

``s2a_level_losses(cls_maps, box_maps, labels, label_weights, bbox_targets, bbox_weights, avg_factor, ...)`` returns a
(2, L) tensor: row 0 the per-level classification losses, row 1 the per-level regression losses -- what
/root/reference/python/jdet/models/roi_heads/s2anet_head.py:430-508 computes with 2 x L loss-module calls over permuted
copies of the maps.  The maps are consumed in place (NCHW, fp32 or bf16); gradients come back in the same layout."""
import ctypes

import torch

from .. import _lib

__all__ = ["s2a_level_losses"]

_ws = {}


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _workspace(dev, nbytes):
    """Zero-on-entry / zero-on-exit counter + partials, one buffer per (device, stream): concurrent use from two
    streams must not share the arrival counter."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    cur = _ws.get(key)
    if cur is None or cur.numel() < nbytes:
        cur = torch.zeros((max(nbytes, 4096) * 2,), dtype=torch.uint8, device=dev)   # counter word zeroed once
        _ws[key] = cur
    return cur


class _S2ALoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, labels, label_weights, bbox_targets, bbox_weights, avg_factor, params, L, *maps):
        lib = _lib.load()
        cls, box = [m.contiguous() for m in maps[:L]], [m.contiguous() for m in maps[L:]]
        dt = cls[0].dtype
        if dt not in (torch.float32, torch.bfloat16) or any(m.dtype != dt for m in cls + box):
            raise _lib.RsdetError("s2a_level_losses: maps must all be float32 or all bfloat16")
        B, C = cls[0].shape[0], cls[0].shape[1]
        hw = (ctypes.c_int * L)(*[int(m.shape[2] * m.shape[3]) for m in cls])
        A = sum(hw)
        assert tuple(labels.shape) == (B, A) and labels.dtype == torch.int32, (labels.shape, labels.dtype, B, A)
        assert all(b.shape[1] == 5 and b.shape[2:] == c.shape[2:] for b, c in zip(box, cls))
        lab, lw = labels.contiguous(), label_weights.contiguous().float()
        bt, bw = bbox_targets.contiguous().float(), bbox_weights.contiguous().float()
        avg = avg_factor.detach().reshape(1).float().contiguous()
        alpha, gamma, beta, w_cls, w_box = params
        out = torch.empty((2, L), dtype=torch.float32, device=lab.device)
        nbytes = lib.rsdet_s2a_loss_ws_size(hw, L, B)
        ws = _workspace(lab.device, nbytes)
        rc = lib.rsdet_s2a_loss_forward(_ptr_array(cls), _ptr_array(box), int(dt == torch.bfloat16), hw, L, B, C,
                                        _lib.ptr(lab), _lib.ptr(lw), _lib.ptr(bt), _lib.ptr(bw), _lib.ptr(avg),
                                        alpha, gamma, beta, w_cls, w_box, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                        _lib.stream_ptr())
        if rc != _lib.RSDET_OK:
            ws.zero_()          # an aborted launch may leave the arrival counter behind
        _lib.check(rc, "rsdet_s2a_loss_forward")
        ctx.save_for_backward(lab, lw, bt, bw, avg, *cls, *box)
        ctx.params, ctx.L, ctx.hw = params, L, hw
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        lab, lw, bt, bw, avg = ctx.saved_tensors[:5]
        L = ctx.L
        cls, box = list(ctx.saved_tensors[5:5 + L]), list(ctx.saved_tensors[5 + L:])
        gcls, gbox = [torch.empty_like(m) for m in cls], [torch.empty_like(m) for m in box]
        alpha, gamma, beta, w_cls, w_box = ctx.params
        go = grad_out.contiguous().float()
        rc = lib.rsdet_s2a_loss_backward(_ptr_array(cls), _ptr_array(box), int(cls[0].dtype == torch.bfloat16), ctx.hw,
                                         L, cls[0].shape[0], cls[0].shape[1], _lib.ptr(lab), _lib.ptr(lw), _lib.ptr(bt),
                                         _lib.ptr(bw), _lib.ptr(avg), _lib.ptr(go), alpha, gamma, beta, w_cls, w_box,
                                         _ptr_array(gcls), _ptr_array(gbox), _lib.stream_ptr())
        _lib.check(rc, "rsdet_s2a_loss_backward")
        return (None,) * 7 + tuple(gcls) + tuple(gbox)


def s2a_level_losses(cls_maps, box_maps, labels, label_weights, bbox_targets, bbox_weights, avg_factor, alpha=0.25,
                     gamma=2.0, beta=1.0 / 9.0, cls_weight=1.0, box_weight=1.0):
    """cls_maps / box_maps: lists (levels) of (B,C,H,W) / (B,5,H,W) CUDA tensors (fp32 or bf16, same type);
    labels (B,A) int32 1-based; label_weights (B,A); bbox_targets / bbox_weights (B,A,5); avg_factor: device scalar.
    -> (2, L) fp32: per-level focal losses, per-level smooth-L1 losses."""
    for m in list(cls_maps) + list(box_maps):
        if not m.is_cuda:
            raise _lib.RsdetError("rs_detection_amd ops run on the GPU only; no CPU fallback")
    L = len(cls_maps)
    assert L == len(box_maps) and 1 <= L <= 8
    if not torch.is_tensor(avg_factor):
        avg_factor = torch.tensor(float(avg_factor), device=labels.device)
    params = (float(alpha), float(gamma), float(beta), float(cls_weight), float(box_weight))
    return _S2ALoss.apply(labels, label_weights, bbox_targets, bbox_weights, avg_factor, params, L, *cls_maps, *box_maps)
