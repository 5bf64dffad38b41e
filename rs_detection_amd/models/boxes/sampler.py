"""PseudoSampler / SamplingResult (/root/reference/python/jdet/models/boxes/sampler.py:6-36,114-130).

Index sets are data-dependent in size, so their SIZES have to come to the host -- once per ``sample`` call: the two
counts travel together (one synchronisation), the index lists themselves are then built with ``nonzero_static`` (size
known, no synchronisation).  The reference's ``nonzero`` + ``unique`` pairs are four synchronisations per call; with two
images and two samplers per Oriented R-CNN step that is what kept the Python thread waiting on the device."""
import torch


def _pos_neg_indices(gt_inds):
    """Ascending int64 indices of ``gt_inds > 0`` and of ``gt_inds == 0`` -- what ``nonzero(...).squeeze(-1).unique()``
    gives for each (nonzero's output is already sorted and duplicate-free) -- with ONE device synchronisation."""
    pos, neg = gt_inds > 0, gt_inds == 0
    if not gt_inds.is_cuda:
        return torch.nonzero(pos).squeeze(-1), torch.nonzero(neg).squeeze(-1)
    npos, nneg = torch.stack([pos.sum(), neg.sum()]).tolist()
    return (torch.nonzero_static(pos, size=int(npos)).squeeze(-1), torch.nonzero_static(neg, size=int(nneg)).squeeze(-1))

from rs_detection_amd.utils.registry import BOXES


class SamplingResult:
    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds].long() - 1
        if gt_bboxes.numel() == 0:
            assert self.pos_assigned_gt_inds.numel() == 0
            self.pos_gt_bboxes = gt_bboxes.new_empty(gt_bboxes.shape).view(-1, max(gt_bboxes.shape[-1], 4))
        else:
            if gt_bboxes.dim() < 2:
                gt_bboxes = gt_bboxes.view(-1, 4)
            self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds, :]
        self.pos_gt_labels = assign_result.labels[pos_inds] if assign_result.labels is not None else None

    @property
    def bboxes(self):
        return torch.cat([self.pos_bboxes, self.neg_bboxes])


@BOXES.register_module()
class PseudoSampler:
    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        pos_inds, neg_inds = _pos_neg_indices(assign_result.gt_inds)
        gt_flags = torch.zeros(bboxes.shape[0], dtype=torch.bool, device=bboxes.device)
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags)


class BaseSampler:
    """sampler.py:39-112: positive/negative sampling with optional gt-as-proposal injection.
    Index sets are data-dependent in size: their two counts come to the host in one synchronisation (module docstring;
    ``jt.nonzero(...).numel()`` in the reference synchronises per list)."""
    box_dim = 4

    def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True, **kwargs):
        self.num, self.pos_fraction = num, pos_fraction
        self.neg_pos_ub, self.add_gt_as_proposals = neg_pos_ub, add_gt_as_proposals

    def sample(self, assign_result, bboxes, gt_bboxes, gt_labels=None, **kwargs):
        gt_bboxes = gt_bboxes.to(torch.float32)
        bboxes = bboxes.to(torch.float32)
        if bboxes.dim() < 2:
            bboxes = bboxes[None, :]
        bboxes = bboxes[:, :self.box_dim]
        gt_flags = torch.zeros(bboxes.shape[0], dtype=torch.bool, device=bboxes.device)
        if self.add_gt_as_proposals:
            bboxes = torch.cat([gt_bboxes, bboxes], dim=0)
            assign_result.add_gt_(gt_labels)
            gt_flags = torch.cat([torch.ones(gt_bboxes.shape[0], dtype=torch.bool, device=bboxes.device), gt_flags])
        num_expected_pos = int(self.num * self.pos_fraction)
        # both candidate lists with one synchronisation (module docstring); the draws below are the reference's
        self._candidates = _pos_neg_indices(assign_result.gt_inds)
        try:
            pos_inds = self._sorted(self._sample_pos(assign_result, num_expected_pos, bboxes=bboxes, **kwargs))
            num_expected_neg = self.num - pos_inds.numel()
            if self.neg_pos_ub >= 0:
                num_expected_neg = min(num_expected_neg, int(self.neg_pos_ub * max(1, pos_inds.numel())))
            neg_inds = self._sorted(self._sample_neg(assign_result, num_expected_neg, bboxes=bboxes, **kwargs))
        finally:
            self._candidates = None
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags)

    @staticmethod
    def _sorted(inds):
        """``inds.unique()`` of the reference (sampler.py:95,100) for index lists that hold no duplicates (a subset of
        nonzero's output): the ascending order, without unique's size synchronisation."""
        return torch.sort(inds)[0] if inds.numel() > 1 else inds


@BOXES.register_module()
class RandomSampler(BaseSampler):
    """sampler.py:132-180.  ``jt.randperm`` is not reproducible across frameworks: compare distributions,
    not indices."""

    @staticmethod
    def random_choice(gallery, num):
        assert len(gallery) >= num
        perm = torch.randperm(gallery.numel(), device=gallery.device)[:num]
        return gallery[perm]

    def _sample_pos(self, assign_result, num_expected, **kwargs):
        cand = getattr(self, "_candidates", None)
        pos_inds = cand[0] if cand is not None else torch.nonzero(assign_result.gt_inds > 0).squeeze(1)
        return pos_inds if pos_inds.numel() <= num_expected else self.random_choice(pos_inds, num_expected)

    def _sample_neg(self, assign_result, num_expected, **kwargs):
        cand = getattr(self, "_candidates", None)
        neg_inds = cand[1] if cand is not None else torch.nonzero(assign_result.gt_inds == 0).squeeze(1)
        return neg_inds if len(neg_inds) <= num_expected else self.random_choice(neg_inds, num_expected)


@BOXES.register_module()
class RandomSamplerRotated(RandomSampler):
    """sampler.py:182-232: identical except proposals keep 5 columns."""
    box_dim = 5
