"""FocalLoss (/root/reference/python/jdet/models/losses/focal_loss.py:36-96).

Sigmoid focal loss on 1-based labels (0 = background, column c <-> class c+1, :37-38);
BCE-with-logits in the reference's max_val form (:5-21) with ``weight`` broadcast over
the class axis; ``reduction='mean'`` divides the SUM by ``avg_factor`` (:49-52)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from rs_detection_amd.utils.registry import LOSSES


def sigmoid_focal_loss(inputs, targets, weight=None, alpha=-1, gamma=2, reduction="none", avg_factor=None):
    C = inputs.shape[1]
    t = (torch.arange(1, C + 1, device=inputs.device, dtype=targets.dtype)[None, :] == targets[:, None]).to(inputs.dtype)
    p = inputs.sigmoid()
    ce = F.binary_cross_entropy_with_logits(inputs, t, reduction="none")
    if weight is not None:
        ce = ce * weight[:, None]
    p_t = p * t + (1 - p) * (1 - t)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * t + (1 - alpha) * (1 - t)) * loss
    if reduction == "mean":
        loss = loss.sum() / (loss.numel() if avg_factor is None else avg_factor)
    elif reduction == "sum":
        loss = loss.sum()
    return loss


@LOSSES.register_module()
class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.use_sigmoid, self.gamma, self.alpha = use_sigmoid, gamma, alpha
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * sigmoid_focal_loss(pred, target, weight, gamma=self.gamma, alpha=self.alpha,
                                                     reduction=reduction, avg_factor=avg_factor)
