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
