"""SmoothL1Loss (/root/reference/python/jdet/models/losses/smooth_l1_loss.py:5-54)."""
import torch
import torch.nn as nn

from rs_detection_amd.utils.registry import LOSSES


def smooth_l1_loss(pred, target, weight=None, beta=1., avg_factor=None, reduction="mean"):
    diff = (pred - target).abs()
    if beta != 0.:
        loss = torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta)
    else:
        loss = diff
    if weight is not None:
        loss = loss * (weight[:, None] if weight.dim() == 1 else weight)
    if avg_factor is None:
        avg_factor = max(loss.shape[0], 1)
    if reduction == "mean":
        loss = loss.sum() / avg_factor
    elif reduction == "sum":
        loss = loss.sum()
    return loss


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * smooth_l1_loss(pred, target, weight, beta=self.beta, reduction=reduction,
                                                 avg_factor=avg_factor)
