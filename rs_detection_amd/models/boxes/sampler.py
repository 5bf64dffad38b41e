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

from rs_detection_amd.ops import orpn
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
    def priorities(n, device):
        """One uniform draw per candidate: "the k candidates with the largest draws" is a uniform k-subset, which is what
        ``gallery[randperm(len(gallery))[:k]]`` (sampler.py:139-149) draws -- without knowing len(gallery) on the host."""
        return torch.rand((n,), device=device)

    def sample_masked(self, assign_result, bboxes, gt_bboxes, gt_labels=None, valid=None):
        """``sample`` (sampler.py:57-111) with FIXED-SIZE outputs and no host synchronisation: the train step's form.
        Same distribution of samples: min(#pos, num * pos_fraction) positives and negatives up to ``num`` (``neg_pos_ub``
        honoured), each a uniform subset.  ``valid`` (n,) bool: rows of ``bboxes`` that are padding (a fixed-size
        proposal list) are neither positive nor negative.  Does not modify ``assign_result``."""
        gt_bboxes, bboxes = gt_bboxes.to(torch.float32), bboxes.to(torch.float32)
        if bboxes.dim() < 2:
            bboxes = bboxes[None, :]
        bboxes = bboxes[:, :self.box_dim]
        dev = bboxes.device
        gt_inds, labels = assign_result.gt_inds, assign_result.labels
        # one radix select on the device (csrc/orpn.hip: 6 launches) instead of two top-k's, an argsort and ~40 small
        # tensor operations; the tensor form below is the CPU path and what tests/test_gpu_orpn.py compares it with
        K = gt_bboxes.shape[0] if self.add_gt_as_proposals else 0
        num = int(self.num)
        r = self.priorities(bboxes.shape[0] + K, dev)
        if orpn.sampler_applies(gt_inds, r, num):
            inds, is_pos, val, assigned, counts = orpn.sample_masked(gt_inds, valid, K, r, num,
                                                                     int(self.num * self.pos_fraction), self.neg_pos_ub)
            if self.add_gt_as_proposals:
                bboxes = torch.cat([gt_bboxes, bboxes], dim=0)
                if labels is not None:
                    labels = torch.cat([gt_labels.to(labels.dtype), labels])
            pos_gt = gt_bboxes[assigned, :] if gt_bboxes.shape[0] > 0 else \
                gt_bboxes.new_zeros((num, max(gt_bboxes.shape[-1], 4)))
            return MaskedSamples(inds=inds, is_pos=is_pos, valid=val, bboxes=bboxes[inds], pos_gt_bboxes=pos_gt,
                                 pos_gt_labels=labels[inds] if labels is not None else None, n_pos=counts[0],
                                 n_neg=counts[1], num_gts=gt_bboxes.shape[0], assigned=assigned)
        if valid is not None:
            gt_inds = torch.where(valid, gt_inds, torch.full_like(gt_inds, -1))
        if self.add_gt_as_proposals:
            K = gt_bboxes.shape[0]
            bboxes = torch.cat([gt_bboxes, bboxes], dim=0)
            gt_inds = torch.cat([torch.arange(1, K + 1, dtype=gt_inds.dtype, device=dev), gt_inds])
            if labels is not None:
                labels = torch.cat([gt_labels.to(labels.dtype), labels])
        n = bboxes.shape[0]
        minus = torch.full_like(r, -1.0)
        kp, kn = min(int(self.num * self.pos_fraction), n), min(num, n)
        pk, pi = torch.topk(torch.where(gt_inds > 0, r, minus), kp)
        pv = pk >= 0
        n_pos = pv.sum()
        quota = num - n_pos
        if self.neg_pos_ub >= 0:
            quota = torch.minimum(quota, (self.neg_pos_ub * n_pos.clamp(min=1)).long())
        nk, ni = torch.topk(torch.where(gt_inds == 0, r, minus), kn)
        nv = (nk >= 0) & (torch.arange(kn, device=dev) < quota)
        inds = torch.cat([pi, ni])
        is_pos = torch.cat([pv, torch.zeros_like(nv)])
        val = torch.cat([pv, nv])
        # the reference's order: positives by ascending index, negatives by ascending index (its `.unique()`), rest last
        key = torch.where(val, (~is_pos).long(), torch.full_like(inds, 2)) * (n + 1) + inds
        order = torch.argsort(key)[:num]
        inds, is_pos, val = inds[order], is_pos[order], val[order]
        if inds.numel() < num:                     # fewer boxes than slots: pad with unused slots
            pad = num - inds.numel()
            inds = torch.cat([inds, inds.new_zeros(pad)])
            is_pos = torch.cat([is_pos, is_pos.new_zeros(pad)])
            val = torch.cat([val, val.new_zeros(pad)])
        is_pos = is_pos & val
        gi = (gt_inds[inds].long() - 1).clamp(min=0)
        if gt_bboxes.shape[0] > 0:
            pos_gt = gt_bboxes[gi, :]
        else:
            pos_gt = gt_bboxes.new_zeros((num, max(gt_bboxes.shape[-1], 4)))
        return MaskedSamples(inds=inds, is_pos=is_pos, valid=val, bboxes=bboxes[inds], pos_gt_bboxes=pos_gt,
                             pos_gt_labels=labels[inds] if labels is not None else None, n_pos=n_pos,
                             n_neg=(val & ~is_pos).sum(), num_gts=gt_bboxes.shape[0], assigned=gi)

    @staticmethod
    def _sorted(inds):
        """``inds.unique()`` of the reference (sampler.py:95,100) for index lists that hold no duplicates (a subset of
        nonzero's output): the ascending order, without unique's size synchronisation."""
        return torch.sort(inds)[0] if inds.numel() > 1 else inds


class MaskedSamples:
    """What ``BaseSampler.sample_masked`` returns: exactly ``num`` rows whatever the data -- the sampled boxes in the
    reference's order (positives by ascending index, then negatives by ascending index), unused slots last -- with
    masks instead of index lists of data-dependent length.  Everything is a device tensor; nothing was synchronised.
      inds (num,) int64 rows of the (gt-extended) box list; is_pos / valid (num,) bool; bboxes (num, d);
      pos_gt_bboxes (num, d) and pos_gt_labels (num,) -- defined where is_pos; n_pos / n_neg 0-d int64; assigned (num,)
      int64 the ground-truth row of a positive (0 elsewhere)."""
    __slots__ = ("inds", "is_pos", "valid", "bboxes", "pos_gt_bboxes", "pos_gt_labels", "n_pos", "n_neg", "num_gts",
                 "assigned")

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


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
