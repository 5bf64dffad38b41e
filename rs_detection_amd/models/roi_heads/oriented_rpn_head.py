"""OrientedRPNHead (/root/reference/python/jdet/models/roi_heads/oriented_rpn_head.py:9-492).

Same constructor, parameters (rpn_conv / rpn_cls / rpn_reg), losses and proposal routine.  gt theta is negated
(:281-282); anchors are horizontal (611 072 per 1024^2 tile with 7 ratios); proposals go through the
hbb NMS with per-level coordinate offsets (:213-219, the role of jt.nms)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from rs_detection_amd.models.boxes.anchor_target import images_to_levels, anchor_inside_flags
from rs_detection_amd.ops import orpn
from rs_detection_amd.ops.bbox_transforms import obb2hbb, get_bbox_type, get_bbox_dim, bbox2type
from rs_detection_amd.ops.nms import nms as hbb_nms
from rs_detection_amd.utils.general import multi_apply
from rs_detection_amd.utils.registry import BOXES, LOSSES, HEADS, build_from_cfg


@HEADS.register_module()
class OrientedRPNHead(nn.Module):
    def __init__(self, in_channels, num_classes=1, min_bbox_size=0, nms_thresh=0.8, nms_pre=2000, nms_post=2000,
                 feat_channels=256, bbox_type='obb', reg_dim=6, background_label=0, reg_decoded_bbox=False,
                 pos_weight=-1,
                 anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0],
                                       strides=[4, 8, 16, 32, 64]),
                 bbox_coder=dict(type='MidpointOffsetCoder', target_means=[.0, .0, .0, .0, .0, .0],
                                 target_stds=[1.0, 1.0, 1.0, 1.0, 0.5, 0.5]),
                 loss_cls=dict(type='CrossEntropyLossForRcnn', use_sigmoid=True, loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
                 assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3,
                               ignore_iof_thr=-1, match_low_quality=True, assigned_labels_filled=-1),
                 sampler=dict(type='RandomSampler', num=256, pos_fraction=0.5, neg_pos_ub=-1,
                              add_gt_as_proposals=False)):
        super().__init__()
        self.min_bbox_size, self.nms_thresh, self.nms_pre, self.nms_post = min_bbox_size, nms_thresh, nms_pre, nms_post
        self.in_channels, self.feat_channels, self.num_classes = in_channels, feat_channels, num_classes
        self.unmap_outputs, self.bbox_type, self.reg_dim, self.pos_weight = True, bbox_type, reg_dim, pos_weight
        self.use_sigmoid_cls = loss_cls.get('use_sigmoid', False)
        self.sampling = loss_cls['type'] not in ['FocalLoss', 'GHMC', 'QualityFocalLoss']
        self.cls_out_channels = num_classes if self.use_sigmoid_cls else num_classes + 1
        self.reg_decoded_bbox = reg_decoded_bbox
        self.background_label = num_classes if background_label is None else background_label
        assert self.background_label == 0 or self.background_label == num_classes
        self.bbox_coder = build_from_cfg(bbox_coder, BOXES)
        self.loss_cls = build_from_cfg(loss_cls, LOSSES)
        self.loss_bbox = build_from_cfg(loss_bbox, LOSSES)
        self.assigner = build_from_cfg(assigner, BOXES)
        self.sampler = build_from_cfg(sampler, BOXES)
        self.anchor_generator = build_from_cfg(anchor_generator, BOXES)
        self.num_anchors = self.anchor_generator.num_base_anchors[0]
        self.masked = True      # train step on fixed-size samples / proposals (no host synchronisation)
        self._init_layers()

    def _init_layers(self):
        self.rpn_conv = nn.Conv2d(self.in_channels, self.feat_channels, 3, padding=1)
        self.rpn_cls = nn.Conv2d(self.feat_channels, self.num_anchors * self.num_classes, 1)
        self.rpn_reg = nn.Conv2d(self.feat_channels, self.num_anchors * 6, 1)

    @staticmethod
    def unmap(data, count, inds, fill=0):
        """``inds``: the bool flags of the reference (anchor_target.py `unmap`) or, from the cached geometry below, the
        int64 indices they select -- the index form needs no device synchronisation (a bool mask does: nonzero)."""
        ret = data.new_full((count,) + tuple(data.shape[1:]), fill)
        if inds.dtype == torch.bool:
            ret[inds] = data
        else:
            ret.index_copy_(0, inds, data)
        return ret

    def _inside_geometry(self, anchors_list, valid_flag_list, img_size):
        """flat anchors, inside flags, their indices and the anchors they select, per (anchor set, valid flags, image
        size): the anchor grid is the same tensor list every step (AnchorGenerator caches it) and the flags depend on
        the tile shape only, so the two synchronising selections `bool(inside.any())` / `flat_anchors[inside]` run once
        instead of every image of every step."""
        key = (tuple(id(a) for a in anchors_list), tuple(id(f) for f in valid_flag_list),
               tuple(int(v) for v in img_size[:2]))
        hit = self._geom_cache.get(key) if hasattr(self, "_geom_cache") else None
        if hit is not None and all(a is b for a, b in zip(hit["anchors_list"], anchors_list)) and \
                all(a is b for a, b in zip(hit["flags"], valid_flag_list)):      # (the ids belong to live tensors)
            return hit
        flat_anchors, valid_flags = torch.cat(anchors_list), torch.cat(valid_flag_list)
        inside = anchor_inside_flags(flat_anchors, valid_flags, img_size[:2], allowed_border=0)
        idx = torch.nonzero(inside).squeeze(1)
        geom = dict(anchors_list=list(anchors_list), flags=list(valid_flag_list), flat=flat_anchors,
                    inside=inside, idx=idx, anchors=flat_anchors[idx, :], any=idx.numel() > 0)
        if not hasattr(self, "_geom_cache"):
            self._geom_cache = {}
        if len(self._geom_cache) > 8:
            self._geom_cache.clear()
        self._geom_cache[key] = geom
        return geom

    def forward_single(self, x):
        c = self.rpn_conv
        if (x.is_cuda and type(c) is nn.Conv2d and c.bias is not None and c.padding_mode == 'zeros'
                and x.dtype == torch.float32 and not torch.is_autocast_enabled()):
            # bias + ReLU as ONE pass behind the convolution (ops/bn_act.bias_act), its backward one pass that also sums the
            # bias gradient -- instead of MIOpen's bias add + clamp forward and threshold + strided reduction backward
            from rs_detection_amd.ops.bn_act import bias_act
            x = bias_act(F.conv2d(x, c.weight, None, c.stride, c.padding, c.dilation, c.groups), c.bias, relu=True)
        else:
            x = F.relu(c(x))
        return self.rpn_cls(x), self.rpn_reg(x)

    def _get_bboxes_single(self, cls_scores, bbox_preds, mlvl_anchors, img_shape, fixed=False):
        """``fixed=True`` (the train step): the same proposals as a FIXED-SIZE list -- (nms_post, 6) rows in the same
        order, zero rows behind the last proposal, plus the (nms_post,) bool mask of the real ones -- built without the
        two host synchronisations of the reference's form (the `valid.all()` test and the boolean `dets[keep]`)."""
        level_ids, mlvl_scores, mlvl_valid_anchors, mlvl_bbox_pred = [], [], [], []
        for idx in range(len(cls_scores)):
            cls, reg = cls_scores[idx], bbox_preds[idx]
            assert cls.shape[-2:] == reg.shape[-2:]
            cls = cls.permute(1, 2, 0)
            scores = cls.reshape(-1).sigmoid() if self.use_sigmoid_cls else cls.reshape(-1, 2).softmax(dim=1)[:, 1]
            reg = reg.permute(1, 2, 0).reshape(-1, self.reg_dim)
            anchors = mlvl_anchors[idx]
            if self.nms_pre > 0 and scores.shape[0] > self.nms_pre:
                ranked, rank_inds = scores.sort(descending=True, stable=True)
                topk = rank_inds[:self.nms_pre]
                scores, reg, anchors = ranked[:self.nms_pre], reg[topk, :], anchors[topk, :]
            mlvl_scores.append(scores)
            mlvl_bbox_pred.append(reg)
            mlvl_valid_anchors.append(anchors)
            level_ids.append(scores.new_full((scores.size(0),), idx, dtype=torch.long))
        anchors, reg, scores = torch.cat(mlvl_valid_anchors), torch.cat(mlvl_bbox_pred), torch.cat(mlvl_scores)
        proposals = self.bbox_coder.decode(anchors, reg.float(), max_shape=img_shape)
        ids = torch.cat(level_ids)
        if fixed:
            return self._nms_fixed(proposals, scores, ids)
        if self.min_bbox_size >= 0:
            valid = (proposals[:, 2] > self.min_bbox_size) & (proposals[:, 3] > self.min_bbox_size)
            if not bool(valid.all()):
                proposals, scores, ids = proposals[valid], scores[valid], ids[valid]
        hprop = obb2hbb(proposals)
        max_coordinate = hprop.max() - hprop.min()
        hprop = hprop + (ids.to(hprop.dtype) * (max_coordinate + 1))[:, None]
        keep = hbb_nms(torch.cat([hprop, scores.unsqueeze(1)], dim=1).float().contiguous(), self.nms_thresh)
        dets = torch.cat([proposals, scores.unsqueeze(1)], dim=1)[keep, :]
        return dets[:self.nms_post]

    def _nms_fixed(self, proposals, scores, ids):
        """The tail of _get_bboxes_single on masks: too-small boxes are sorted behind every real one (score -1) and
        never kept, the NMS keep flags (score order) are turned into output slots by a prefix count, and the kept rows
        are scattered into a (nms_post + 1)-row buffer whose last row collects everything that is not an output."""
        from rs_detection_amd import _lib
        n, dev = proposals.shape[0], proposals.device
        ok = torch.ones((n,), dtype=torch.bool, device=dev)
        if self.min_bbox_size >= 0:
            ok = (proposals[:, 2] > self.min_bbox_size) & (proposals[:, 3] > self.min_bbox_size)
        hprop = obb2hbb(proposals)
        inf = torch.full_like(hprop, float("inf"))                                 # the extent of the REAL boxes
        max_coordinate = torch.where(ok[:, None], hprop, -inf).max() - torch.where(ok[:, None], hprop, inf).min()
        hprop = hprop + (ids.to(hprop.dtype) * (max_coordinate + 1))[:, None]
        key = torch.where(ok, scores, torch.full_like(scores, -1.0))
        order = torch.argsort(key, descending=True, stable=True)
        boxes = hprop[order].float().contiguous()
        keep = torch.empty((n,), dtype=torch.uint8, device=dev)
        lib = _lib.load()
        ws_bytes = lib.rsdet_nms_hbb_ws_size(n)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
        rc = lib.rsdet_nms_hbb_sorted_f32(_lib.ptr(boxes), n, float(self.nms_thresh), 1, _lib.ptr(keep), _lib.ptr(ws),
                                          ws_bytes, _lib.stream_ptr())
        _lib.check(rc, "rsdet_nms_hbb_sorted_f32")
        kept = keep.bool() & ok[order]
        slot = torch.cumsum(kept, 0) - 1
        P = int(self.nms_post)
        slot = torch.where(kept & (slot < P), slot, torch.full_like(slot, P))
        dets = torch.cat([proposals, scores.unsqueeze(1)], dim=1)[order]
        out = dets.new_zeros((P + 1, dets.shape[1]))
        out.index_copy_(0, slot, dets)          # (row P receives the discards, in no particular order)
        flags = torch.zeros((P + 1,), dtype=torch.bool, device=dev)
        flags.index_fill_(0, slot, True)
        return out[:P], flags[:P]

    def get_bboxes(self, cls_scores, bbox_preds, targets, fixed=False):
        assert len(cls_scores) == len(bbox_preds)
        num_levels = len(cls_scores)
        featmap_sizes = [tuple(cls_scores[i].shape[-2:]) for i in range(num_levels)]
        mlvl_anchors = self.anchor_generator.grid_anchors(featmap_sizes, device=cls_scores[0].device)
        coder = self.bbox_coder
        if fixed and self.use_sigmoid_cls and self.cls_out_channels == 1 and type(coder).__name__ == "MidpointOffsetCoder" \
                and orpn.proposals_apply(cls_scores, bbox_preds, mlvl_anchors, self.nms_pre, self.nms_post):
            # the whole batch in 12 launches (csrc/orpn.hip); _get_bboxes_single below is the same routine image by image
            with torch.no_grad():
                dets, real = orpn.proposals([orpn.pixel_major_sigmoid(c.detach()) for c in cls_scores], [r.detach() for r in bbox_preds],
                                           mlvl_anchors, self.nms_pre, self.nms_post, self.nms_thresh, self.min_bbox_size,
                                           coder.means, coder.stds, abs(math.log(16 / 1000)))
            return [(dets[i], real[i]) for i in range(len(targets))]
        out = []
        for img_id, target in enumerate(targets):
            cls_list = [cls_scores[i][img_id].detach() for i in range(num_levels)]
            reg_list = [bbox_preds[i][img_id].detach() for i in range(num_levels)]
            out.append(self._get_bboxes_single(cls_list, reg_list, mlvl_anchors, target['img_size'], fixed=fixed))
        return out

    def _get_targets_single(self, anchors_list, valid_flag_list, target):
        dev = anchors_list[0].device
        gt_bboxes = torch.as_tensor(target["rboxes"]).to(dev).float().clone()
        gt_bboxes[:, -1] *= -1
        ign = target.get("rboxes_ignore")
        gt_ignore = None
        if ign is not None and torch.as_tensor(ign).numel() > 0:
            gt_ignore = torch.as_tensor(ign).to(dev).float().clone()
            gt_ignore[:, -1] *= -1
        gt_labels = None
        geom = self._inside_geometry(anchors_list, valid_flag_list, target["img_size"])
        if not geom["any"]:
            return (None,) * 7
        flat_anchors, inside, anchors = geom["flat"], geom["idx"], geom["anchors"]
        a_type, g_type = get_bbox_type(anchors), get_bbox_type(gt_bboxes)
        tgt = bbox2type(gt_bboxes, a_type)
        tgt_ign = None if gt_ignore is None else bbox2type(gt_ignore, a_type)
        assign_result = self.assigner.assign(anchors, tgt, tgt_ign, None if self.sampling else gt_labels)
        sampling_result = self.sampler.sample(assign_result, anchors, tgt)
        if a_type != g_type:
            if gt_bboxes.numel() == 0:
                sampling_result.pos_gt_bboxes = gt_bboxes.new_empty((0, get_bbox_dim(g_type)))
            else:
                sampling_result.pos_gt_bboxes = gt_bboxes[sampling_result.pos_assigned_gt_inds, :]
        n = anchors.shape[0]
        bbox_targets = anchors.new_zeros((n, self.reg_dim))
        bbox_weights = anchors.new_zeros((n, self.reg_dim))
        labels = anchors.new_full((n,), self.background_label, dtype=torch.long)
        label_weights = anchors.new_zeros((n,))
        pos_inds, neg_inds = sampling_result.pos_inds, sampling_result.neg_inds
        if len(pos_inds) > 0:
            pos_t = sampling_result.pos_gt_bboxes if self.reg_decoded_bbox else \
                self.bbox_coder.encode(sampling_result.pos_bboxes, sampling_result.pos_gt_bboxes)
            # (index_fill_ for the constants: `t[idx] = 1.0` builds the value on the host and copies it over, one device
            #  synchronisation per statement)
            bbox_targets[pos_inds, :] = pos_t
            bbox_weights.index_fill_(0, pos_inds, 1.0)
            labels.index_fill_(0, pos_inds, 1)  # only the RPN passes gt_labels=None: FG is 1 (:323-325)
            label_weights.index_fill_(0, pos_inds, 1.0 if self.pos_weight <= 0 else float(self.pos_weight))
        if len(neg_inds) > 0:
            label_weights.index_fill_(0, neg_inds, 1.0)
        if self.unmap_outputs:
            total = flat_anchors.size(0)
            labels = self.unmap(labels, total, inside, fill=self.background_label)
            label_weights = self.unmap(label_weights, total, inside)
            bbox_targets = self.unmap(bbox_targets, total, inside)
            bbox_weights = self.unmap(bbox_weights, total, inside)
        return labels, label_weights, bbox_targets, bbox_weights, pos_inds, neg_inds, sampling_result

    def _get_targets_single_masked(self, anchors_list, valid_flag_list, target):
        """_get_targets_single on fixed-size samples (sampler.sample_masked): the same target maps, no host
        synchronisation.  Index lists become (index, mask) pairs; a masked-out slot writes into one spare row behind the
        maps.  Returns (labels, label_weights, bbox_targets, bbox_weights, n_pos, n_neg) with 0-d device counts."""
        dev = anchors_list[0].device
        gt_bboxes = torch.as_tensor(target["rboxes"]).to(dev).float().clone()
        gt_bboxes[:, -1] *= -1
        ign = target.get("rboxes_ignore")
        gt_ignore = None
        if ign is not None and torch.as_tensor(ign).numel() > 0:
            gt_ignore = torch.as_tensor(ign).to(dev).float().clone()
            gt_ignore[:, -1] *= -1
        geom = self._inside_geometry(anchors_list, valid_flag_list, target["img_size"])
        if not geom["any"]:
            return (None,) * 6
        flat_anchors, inside, anchors = geom["flat"], geom["idx"], geom["anchors"]
        a_type, g_type = get_bbox_type(anchors), get_bbox_type(gt_bboxes)
        tgt = bbox2type(gt_bboxes, a_type)
        tgt_ign = None if gt_ignore is None else bbox2type(gt_ignore, a_type)
        assign_result = self.assigner.assign(anchors, tgt, tgt_ign, None)
        ms = self.sampler.sample_masked(assign_result, anchors, tgt)
        pos_gt = ms.pos_gt_bboxes
        if a_type != g_type:
            gi = (assign_result.gt_inds[ms.inds].long() - 1).clamp(min=0)
            pos_gt = gt_bboxes[gi, :] if gt_bboxes.numel() else gt_bboxes.new_zeros((ms.inds.numel(), get_bbox_dim(g_type)))
        n = anchors.shape[0]
        at_pos = torch.where(ms.is_pos, ms.inds, torch.full_like(ms.inds, n))       # spare row n: masked-out slots
        at_any = torch.where(ms.valid, ms.inds, torch.full_like(ms.inds, n))
        pos_t = pos_gt if self.reg_decoded_bbox else self.bbox_coder.encode(ms.bboxes, pos_gt)
        pos_t = torch.where(ms.is_pos[:, None], pos_t, torch.zeros_like(pos_t))
        bbox_targets = anchors.new_zeros((n + 1, self.reg_dim)).index_copy_(0, at_pos, pos_t)[:n]
        bbox_weights = anchors.new_zeros((n + 1, self.reg_dim)).index_fill_(0, at_pos, 1.0)[:n]
        labels = anchors.new_full((n + 1,), self.background_label, dtype=torch.long).index_fill_(0, at_pos, 1)[:n]
        label_weights = anchors.new_zeros((n + 1,)).index_fill_(0, at_any, 1.0)
        if self.pos_weight > 0:
            label_weights.index_fill_(0, at_pos, float(self.pos_weight))
        label_weights = label_weights[:n]
        if self.unmap_outputs:
            total = flat_anchors.size(0)
            labels = self.unmap(labels, total, inside, fill=self.background_label)
            label_weights = self.unmap(label_weights, total, inside)
            bbox_targets = self.unmap(bbox_targets, total, inside)
            bbox_weights = self.unmap(bbox_weights, total, inside)
        return labels, label_weights, bbox_targets, bbox_weights, ms.n_pos, ms.n_neg

    def get_targets_masked(self, anchor_list, valid_flag_list, targets):
        """get_targets with device-side counts: num_total_pos / num_total_neg are 0-d tensors (sum_img max(count, 1))."""
        num_level_anchors = [a.size(0) for a in anchor_list[0]]
        (all_labels, all_lw, all_bt, all_bw, npos, nneg) = multi_apply(self._get_targets_single_masked, anchor_list,
                                                                         valid_flag_list, targets)
        num_total_pos = torch.stack([c.clamp(min=1) for c in npos]).sum()
        num_total_neg = torch.stack([c.clamp(min=1) for c in nneg]).sum()
        return (images_to_levels(all_labels, num_level_anchors), images_to_levels(all_lw, num_level_anchors),
                images_to_levels(all_bt, num_level_anchors), images_to_levels(all_bw, num_level_anchors),
                num_total_pos, num_total_neg)

    def get_targets(self, anchor_list, valid_flag_list, targets):
        num_level_anchors = [a.size(0) for a in anchor_list[0]]
        (all_labels, all_lw, all_bt, all_bw, pos_l, neg_l, _) = multi_apply(self._get_targets_single, anchor_list,
                                                                             valid_flag_list, targets)
        num_total_pos = sum(max(i.numel(), 1) for i in pos_l)
        num_total_neg = sum(max(i.numel(), 1) for i in neg_l)
        return (images_to_levels(all_labels, num_level_anchors), images_to_levels(all_lw, num_level_anchors),
                images_to_levels(all_bt, num_level_anchors), images_to_levels(all_bw, num_level_anchors),
                num_total_pos, num_total_neg)

    def loss_single(self, cls_score, bbox_pred, anchors, labels, label_weights, bbox_targets, bbox_weights,
                    num_total_samples):
        labels, label_weights = labels.reshape(-1), label_weights.reshape(-1)
        cls_score = cls_score.permute(0, 2, 3, 1).reshape(-1, self.cls_out_channels)
        loss_cls = self.loss_cls(cls_score, labels, label_weights, avg_factor=num_total_samples)
        bbox_targets, bbox_weights = bbox_targets.reshape(-1, self.reg_dim), bbox_weights.reshape(-1, self.reg_dim)
        bbox_pred = bbox_pred.permute(0, 2, 3, 1).reshape(-1, self.reg_dim)
        if self.reg_decoded_bbox:
            bbox_pred = self.bbox_coder.decode(anchors.reshape(-1, anchors.size(-1)), bbox_pred)
        loss_bbox = self.loss_bbox(bbox_pred, bbox_targets, bbox_weights, avg_factor=num_total_samples)
        return loss_cls, loss_bbox

    def _valid_flags(self, featmap_sizes, pad_shape, dev):
        """AnchorGenerator.valid_flags per (pyramid shape, padded tile shape): the same tensors every step (they are the
        key of the cached inside-anchor geometry above, and ~30 small launches per image otherwise)."""
        key = (tuple(featmap_sizes), tuple(int(v) for v in pad_shape[:2]), str(dev))
        cache = self.__dict__.setdefault("_vf_cache", {})
        if key not in cache:
            if len(cache) > 8:
                cache.clear()
            cache[key] = self.anchor_generator.valid_flags(featmap_sizes, pad_shape, device=dev)
        return cache[key]

    def loss(self, cls_scores, bbox_preds, targets):
        featmap_sizes = [tuple(f.shape[-2:]) for f in cls_scores]
        assert len(featmap_sizes) == self.anchor_generator.num_levels
        dev = cls_scores[0].device
        mla = self.anchor_generator.grid_anchors(featmap_sizes, device=dev)
        anchor_list = [mla for _ in range(len(targets))]
        valid_flag_list = [self._valid_flags(featmap_sizes, t['pad_shape'], dev) for t in targets]
        if self.masked:
            sparse = self._loss_on_samples(cls_scores, bbox_preds, anchor_list, valid_flag_list, targets)
            if sparse is not None:
                return sparse
        # the train step's form: fixed-size samples, counts on the device (no host synchronisation); `masked = False`
        # keeps the reference-shaped index lists (what the known-answer tests of the head read)
        get = self.get_targets_masked if self.masked else self.get_targets
        (labels_list, lw_list, bt_list, bw_list, npos, nneg) = get(anchor_list, valid_flag_list, targets)
        num_level_anchors = [a.size(0) for a in anchor_list[0]]
        all_anchor_list = images_to_levels([torch.cat(a) for a in anchor_list], num_level_anchors)
        losses_cls, losses_bbox = multi_apply(self.loss_single, cls_scores, bbox_preds, all_anchor_list, labels_list,
                                              lw_list, bt_list, bw_list, num_total_samples=npos + nneg)
        return dict(loss_rpn_cls=losses_cls, loss_rpn_bbox=losses_bbox)

    def _loss_on_samples(self, cls_scores, bbox_preds, anchor_list, valid_flag_list, targets):
        """Both losses evaluated on the <= `num` sampled anchors of each image (csrc/orpn.hip: the dense target maps of
        get_targets_masked are zero-weighted everywhere else, so the per-level sums are the same): assignment and one
        radix-select sampler call per image, then ONE launch for all images and levels, and one in the backward.  None
        when the configuration is not the one the kernel restates (then the dense route below runs)."""
        from rs_detection_amd.models.boxes.sampler import RandomSampler
        from rs_detection_amd.models.losses.cross_entropy_loss import CrossEntropyLossForRcnn
        from rs_detection_amd.models.losses.smooth_l1_loss import SmoothL1Loss
        sampler, coder, lc, lb = self.sampler, self.bbox_coder, self.loss_cls, self.loss_bbox
        N, num = len(targets), int(sampler.num)
        if not (isinstance(sampler, RandomSampler) and not sampler.add_gt_as_proposals
                and type(coder).__name__ == "MidpointOffsetCoder" and self.use_sigmoid_cls and self.cls_out_channels == 1
                and self.reg_dim == 6 and not self.reg_decoded_bbox and self.sampling and self.unmap_outputs
                and type(lc) is CrossEntropyLossForRcnn and lc.use_sigmoid and type(lb) is SmoothL1Loss and lb.beta > 0
                and lb.reduction == "mean" and orpn.rpn_loss_applies(cls_scores, bbox_preds, N, num)):
            return None
        dev = cls_scores[0].device
        geom = [self._inside_geometry(anchor_list[i], valid_flag_list[i], t["img_size"]) for i, t in enumerate(targets)]
        if not all(g is geom[0] and g["any"] for g in geom):
            return None
        anchors = geom[0]["anchors"]
        if get_bbox_type(anchors) != 'hbb':
            return None
        inds = torch.empty((N, num), dtype=torch.int64, device=dev)
        assigned = torch.empty((N, num), dtype=torch.int64, device=dev)
        flags = torch.empty((2, N, num), dtype=torch.bool, device=dev)
        counts = torch.empty((N, 2), dtype=torch.int64, device=dev)
        gts = []
        for i, t in enumerate(targets):
            gt = torch.as_tensor(t["rboxes"]).to(dev).float().clone()
            gt[:, -1] *= -1
            ign = t.get("rboxes_ignore")
            gt_ignore = None
            if ign is not None and torch.as_tensor(ign).numel() > 0:
                gt_ignore = torch.as_tensor(ign).to(dev).float().clone()
                gt_ignore[:, -1] *= -1
            if get_bbox_type(gt) != 'obb':
                return None
            ar = self.assigner.assign(anchors, bbox2type(gt, 'hbb'), None if gt_ignore is None else bbox2type(gt_ignore, 'hbb'),
                                      None)
            pri = sampler.priorities(anchors.shape[0], dev)
            if not orpn.sampler_applies(ar.gt_inds, pri, num):
                return None
            orpn.sample_masked(ar.gt_inds, None, 0, pri, num, int(sampler.num * sampler.pos_fraction), sampler.neg_pos_ub,
                               out=(inds[i], flags[0, i], flags[1, i], assigned[i], counts[i]))
            gts.append(gt.contiguous())
        spec = orpn.RpnLossSpec(anchors=geom[0]["flat"], inside=geom[0]["idx"], gts=gts, inds=inds, is_pos=flags[0],
                                val=flags[1], assigned=assigned, counts=counts, means=coder.means, stds=coder.stds,
                                beta=lb.beta, w_cls=lc.loss_weight, w_box=lb.loss_weight, pos_weight=self.pos_weight)
        losses_cls, losses_bbox = orpn.rpn_loss(spec, list(cls_scores), list(bbox_preds))
        return dict(loss_rpn_cls=losses_cls, loss_rpn_bbox=losses_bbox)

    def forward(self, features, targets):
        """Training: (fixed-size proposals, their masks) + losses -- see _get_bboxes_single(fixed=True); evaluation:
        the reference's variable-length proposal lists."""
        outs = multi_apply(self.forward_single, features)
        losses = self.loss(*outs, targets) if self.training else dict()
        return self.get_bboxes(*outs, targets, fixed=self.training and self.masked), losses
