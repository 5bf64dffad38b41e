"""CrossEntropyLoss / CrossEntropyLossForRcnn
(/root/reference/python/jdet/models/losses/cross_entropy_loss.py:6-58, :60-155)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from rs_detection_amd.utils.registry import LOSSES


def weighted_cross_entropy(pred, label, weight, avg_factor=None, reduce=True):
    if avg_factor is None:
        avg_factor = (weight > 0).sum().float().clamp(min=1.)
    raw = F.cross_entropy(pred, label.long(), reduction='none')
    return (raw * weight).sum()[None] / avg_factor if reduce else raw * weight / avg_factor


def _expand_binary_labels(labels, label_weights, label_channels):
    bin_labels = torch.zeros((labels.size(0), label_channels), dtype=torch.float32, device=labels.device)
    cols = (labels.long() - 1).clamp(min=0)
    bin_labels.scatter_(1, cols[:, None], (labels >= 1).float()[:, None])  # :16-22, no host sync
    return bin_labels, label_weights.view(-1, 1).expand(label_weights.size(0), label_channels)


def weighted_binary_cross_entropy(pred, label, weight, avg_factor=None, **kwargs):
    if pred.dim() != label.dim():
        label, weight = _expand_binary_labels(label, weight, pred.size(-1))
    if avg_factor is None:
        avg_factor = (weight > 0).sum().float().clamp(min=1.)
    return F.binary_cross_entropy_with_logits(pred, label.float(), weight.float(), reduction='sum') / avg_factor


@LOSSES.register_module()
class CrossEntropyLossForRcnn(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, loss_weight=1.0):
        super().__init__()
        assert (use_sigmoid is False) or (use_mask is False)
        if use_mask:
            raise NotImplementedError
        self.use_sigmoid, self.use_mask, self.loss_weight = use_sigmoid, use_mask, loss_weight
        self.cls_criterion = weighted_binary_cross_entropy if use_sigmoid else weighted_cross_entropy

    def forward(self, cls_score, label, label_weight, *args, **kwargs):
        return self.loss_weight * self.cls_criterion(cls_score, label, label_weight, *args, **kwargs)


def cross_entropy_loss(pred, target, weight=None, avg_factor=None, reduction="mean"):
    """:60-83: log-sum-exp form with ``safe_log``; one-hot via index compare."""
    loss = F.cross_entropy(pred, target.reshape(-1).long(), reduction='none')
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        avg_factor = max(loss.shape[0], 1)
    if reduction == "mean":
        loss = loss.sum() / avg_factor
    elif reduction == "sum":
        loss = loss.sum()
    return loss


def binary_cross_entropy_loss(pred, label, weight=None, reduction='mean', avg_factor=None, class_weight=None):
    assert pred.dim() == label.dim() and class_weight is None
    loss = F.binary_cross_entropy_with_logits(pred, label.float(), reduction='none')
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        avg_factor = max(loss.shape[0], 1)
    if reduction == "mean":
        loss = loss.sum() / avg_factor
    elif reduction == "sum":
        loss = loss.sum()
    return loss


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, reduction='mean', use_bce=False, loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight, self.use_bce = reduction, loss_weight, use_bce

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        fn = binary_cross_entropy_loss if self.use_bce else cross_entropy_loss
        return self.loss_weight * fn(pred, target, weight, reduction=reduction, avg_factor=avg_factor)
