"""OrientedHead (/root/reference/python/jdet/models/roi_heads/oriented_head.py:48-624): Oriented R-CNN RoI head.

Per image: MaxIoUAssigner (rotated IoU, v1 convention, csrc/box_iou_rotated.hip) on (K x <=2000) proposals,
RandomSamplerRotated(512, 0.25, gt as proposals), rotated RoIAlign through OrientedSingleRoIExtractor
(csrc/rroi_align.hip), 2 shared FCs, fc_cls (num_classes+1, background LAST) and fc_reg (5, class agnostic),
OrientedDeltaXYWHTCoder targets, CE + SmoothL1.  gt theta is negated and labels are 0-based (:551-552,:564, q19).
No NMS inside the model at test time: ``get_results`` only thresholds (:279-305)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from rs_detection_amd.models.utils.modules import ConvModule
from rs_detection_amd.ops.bbox_transforms import get_bbox_dim, obb2poly
from rs_detection_amd.utils.general import multi_apply
from rs_detection_amd.utils.registry import HEADS, BOXES, LOSSES, ROI_EXTRACTORS, build_from_cfg


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


@HEADS.register_module()
class OrientedHead(nn.Module):
    def __init__(self, num_classes=15, in_channels=256, num_shared_convs=0, num_shared_fcs=2, num_cls_convs=0,
                 num_cls_fcs=0, num_reg_convs=0, num_reg_fcs=0, fc_out_channels=1024, conv_out_channels=256,
                 score_thresh=0.05,
                 assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5,
                               ignore_iof_thr=-1, match_low_quality=False, assigned_labels_filled=-1,
                               iou_calculator=dict(type='BboxOverlaps2D_rotated_v1')),
                 sampler=dict(type='RandomSamplerRotated', num=512, pos_fraction=0.25, neg_pos_ub=-1,
                              add_gt_as_proposals=True),
                 bbox_coder=dict(type='OrientedDeltaXYWHTCoder', target_means=[0., 0., 0., 0., 0.],
                                 target_stds=[0.1, 0.1, 0.2, 0.2, 0.1]),
                 bbox_roi_extractor=dict(type='OrientedSingleRoIExtractor',
                                         roi_layer=dict(type='ROIAlignRotated_v1', output_size=7, sampling_ratio=2),
                                         out_channels=256, extend_factor=(1.4, 1.2), featmap_strides=[4, 8, 16, 32]),
                 loss_cls=dict(type='CrossEntropyLoss'),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0),
                 with_bbox=True, with_shared_head=False, with_avg_pool=False, with_cls=True, with_reg=True,
                 start_bbox_type='obb', end_bbox_type='obb', reg_dim=None, reg_class_agnostic=True,
                 reg_decoded_bbox=False, pos_weight=-1):
        super().__init__()
        assert with_cls or with_reg
        self.with_avg_pool, self.with_cls, self.with_reg, self.with_bbox = with_avg_pool, with_cls, with_reg, with_bbox
        self.with_shared_head, self.in_channels, self.num_classes = with_shared_head, in_channels, num_classes
        self.reg_class_agnostic, self.reg_decoded_bbox = reg_class_agnostic, reg_decoded_bbox
        self.pos_weight, self.score_thresh = pos_weight, score_thresh
        self.use_delta_and_encode_loss = ['kfiou']
        self.roi_feat_size = _pair(7)
        self.roi_feat_area = self.roi_feat_size[0] * self.roi_feat_size[1]
        self.start_bbox_type, self.end_bbox_type = start_bbox_type, end_bbox_type
        assert start_bbox_type in ['hbb', 'obb', 'poly'] and end_bbox_type in ['hbb', 'obb', 'poly']
        self.reg_dim = get_bbox_dim(end_bbox_type) if reg_dim is None else reg_dim
        assert (num_shared_convs + num_shared_fcs + num_cls_convs + num_cls_fcs + num_reg_convs + num_reg_fcs > 0)
        if num_cls_convs > 0 or num_reg_convs > 0:
            assert num_shared_fcs == 0
        if not with_cls:
            assert num_cls_convs == 0 and num_cls_fcs == 0
        if not with_reg:
            assert num_reg_convs == 0 and num_reg_fcs == 0
        self.num_shared_convs, self.num_shared_fcs = num_shared_convs, num_shared_fcs
        self.num_cls_convs, self.num_cls_fcs, self.num_reg_convs, self.num_reg_fcs = \
            num_cls_convs, num_cls_fcs, num_reg_convs, num_reg_fcs
        self.conv_out_channels, self.fc_out_channels = conv_out_channels, fc_out_channels
        self.bbox_coder = build_from_cfg(bbox_coder, BOXES)
        self.loss_cls = build_from_cfg(loss_cls, LOSSES)
        self.loss_bbox = build_from_cfg(loss_bbox, LOSSES)
        self.assigner = build_from_cfg(assigner, BOXES)
        self.sampler = build_from_cfg(sampler, BOXES)
        self.bbox_roi_extractor = build_from_cfg(bbox_roi_extractor, ROI_EXTRACTORS)
        if with_avg_pool:
            self.avg_pool = nn.AvgPool2d(self.roi_feat_size)
        self._init_layers()
        self.init_weights()

    @property
    def custom_cls_channels(self):
        return getattr(self.loss_cls, 'custom_cls_channels', False)

    def _add_conv_fc_branch(self, num_convs, num_fcs, in_channels, is_shared=False):
        last = in_channels
        convs = nn.ModuleList()
        for i in range(num_convs):
            convs.append(ConvModule(last if i == 0 else self.conv_out_channels, self.conv_out_channels, 3, padding=1))
        if num_convs > 0:
            last = self.conv_out_channels
        fcs = nn.ModuleList()
        if num_fcs > 0:
            if (is_shared or self.num_shared_fcs == 0) and not self.with_avg_pool:
                last *= self.roi_feat_area
            for i in range(num_fcs):
                fcs.append(nn.Linear(last if i == 0 else self.fc_out_channels, self.fc_out_channels))
            last = self.fc_out_channels
        return convs, fcs, last

    def _init_layers(self):
        self.shared_convs, self.shared_fcs, last = self._add_conv_fc_branch(self.num_shared_convs, self.num_shared_fcs,
                                                                            self.in_channels, True)
        self.shared_out_channels = last
        self.cls_convs, self.cls_fcs, self.cls_last_dim = self._add_conv_fc_branch(self.num_cls_convs, self.num_cls_fcs, last)
        self.reg_convs, self.reg_fcs, self.reg_last_dim = self._add_conv_fc_branch(self.num_reg_convs, self.num_reg_fcs, last)
        if self.num_shared_fcs == 0 and not self.with_avg_pool:
            if self.num_cls_fcs == 0:
                self.cls_last_dim *= self.roi_feat_area
            if self.num_reg_fcs == 0:
                self.reg_last_dim *= self.roi_feat_area
        self.relu = nn.ReLU(inplace=True)
        if self.with_cls:
            self.fc_cls = nn.Linear(self.cls_last_dim, self.num_classes + 1)
        if self.with_reg:
            self.fc_reg = nn.Linear(self.reg_last_dim, self.reg_dim if self.reg_class_agnostic
                                    else self.reg_dim * self.num_classes)

    def init_weights(self):
        if self.with_cls:
            nn.init.normal_(self.fc_cls.weight, 0, 0.01)
            nn.init.constant_(self.fc_cls.bias, 0)
        if self.with_reg:
            nn.init.normal_(self.fc_reg.weight, 0, 0.001)
            nn.init.constant_(self.fc_reg.bias, 0)
        for ml in (self.shared_fcs, self.cls_fcs, self.reg_fcs):
            for m in ml.modules():
                if isinstance(m, nn.Linear):
                    nn.init.xavier_uniform_(m.weight)
                    nn.init.constant_(m.bias, 0)

    def arb2roi(self, bbox_list, bbox_type='hbb'):
        assert bbox_type in ['hbb', 'obb', 'poly']
        dim = get_bbox_dim(bbox_type)
        rois = []
        for img_id, b in enumerate(bbox_list):
            if b.size(0) > 0:
                rois.append(torch.cat([b.new_full((b.size(0), 1), img_id), b[:, :dim]], dim=-1))
            else:
                rois.append(b.new_zeros((0, dim + 1)))
        return torch.cat(rois, 0)

    def get_results(self, multi_bboxes, multi_scores, score_factors=None, bbox_type='hbb'):
        dim = get_bbox_dim(bbox_type)
        num_classes = multi_scores.size(1) - 1
        if multi_bboxes.shape[1] > dim:
            bboxes = multi_bboxes.view(multi_scores.size(0), -1, dim)
        else:
            bboxes = multi_bboxes[:, None].expand(-1, num_classes, dim)
        scores = multi_scores[:, :-1]
        valid = scores > self.score_thresh
        bboxes = bboxes[valid]
        if score_factors is not None:
            scores = scores * score_factors[:, None]
        scores = scores[valid]
        labels = valid.nonzero()[:, 1]
        if bboxes.numel() == 0:
            return multi_bboxes.new_zeros((0, 9)), multi_bboxes.new_zeros((0,), dtype=torch.int64)
        return torch.cat([obb2poly(bboxes), scores.unsqueeze(1)], dim=1), labels

    def forward_single(self, x, sampling_results, test=False):
        if test is None:                        # ready-made (R, 1 + dim) RoIs
            rois = sampling_results
        elif test:
            rois = self.arb2roi(sampling_results, bbox_type=self.start_bbox_type)
        else:
            rois = self.arb2roi([r.bboxes for r in sampling_results], bbox_type=self.start_bbox_type)
        x = self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rois.float().contiguous())
        for conv in self.shared_convs:
            x = conv(x)
        if self.num_shared_fcs > 0:
            if self.with_avg_pool:
                x = self.avg_pool(x)
            x = x.flatten(1)
            for fc in self.shared_fcs:
                x = F.relu(fc(x))
        x_cls = x_reg = x
        for conv in self.cls_convs:
            x_cls = conv(x_cls)
        if x_cls.dim() > 2:
            x_cls = (self.avg_pool(x_cls) if self.with_avg_pool else x_cls).flatten(1)
        for fc in self.cls_fcs:
            x_cls = F.relu(fc(x_cls))
        for conv in self.reg_convs:
            x_reg = conv(x_reg)
        if x_reg.dim() > 2:
            x_reg = (self.avg_pool(x_reg) if self.with_avg_pool else x_reg).flatten(1)
        for fc in self.reg_fcs:
            x_reg = F.relu(fc(x_reg))
        return (self.fc_cls(x_cls) if self.with_cls else None), (self.fc_reg(x_reg) if self.with_reg else None), rois

    def loss(self, cls_score, bbox_pred, rois, labels, label_weights, bbox_targets, bbox_targets_decode, bbox_weights,
             reduction_override=None, num_samples=None):
        """``num_samples``: the number of sampled RoIs when ``rois`` is a fixed-size list with unused slots (0-d tensor);
        None = every row is a sample (the reference's `bbox_targets.size(0)`, oriented_head.py:421)."""
        losses = dict()
        if cls_score is not None and cls_score.numel() > 0:
            avg = (label_weights > 0).sum().float().clamp(min=1.)
            losses['loss_cls'] = self.loss_cls(cls_score, labels, label_weights, avg_factor=avg,
                                               reduction_override=reduction_override)
        if bbox_pred is not None:
            pos = (labels >= 0) & (labels < self.num_classes)
            if self.reg_decoded_bbox:
                bbox_pred = self.bbox_coder.decode(rois[:, 1:], bbox_pred)
            # masked form of the reference's boolean gather (:392-435): same sum, no host sync
            if self.reg_class_agnostic:
                pred = bbox_pred.view(bbox_pred.size(0), self.reg_dim)
            else:
                idx = labels.clamp(min=0, max=self.num_classes - 1).long()
                pred = bbox_pred.view(bbox_pred.size(0), -1, self.reg_dim)[torch.arange(bbox_pred.size(0)), idx]
            w = bbox_weights * pos[:, None].to(bbox_weights.dtype)
            losses['orcnn_bbox_loss'] = self.loss_bbox(pred, bbox_targets, w,
                                                       avg_factor=bbox_targets.size(0) if num_samples is None else num_samples,
                                                       reduction_override=reduction_override)
        return losses

    def get_bboxes_target_single(self, pos_bboxes, neg_bboxes, pos_gt_bboxes, pos_gt_labels, use_delta_and_decode=False):
        num_pos, num_neg = pos_bboxes.size(0), neg_bboxes.size(0)
        n = num_pos + num_neg
        labels = pos_bboxes.new_full((n,), self.num_classes, dtype=torch.long)
        label_weights = pos_bboxes.new_zeros((n,))
        bbox_targets = pos_bboxes.new_zeros((n, self.reg_dim))
        bbox_weights = pos_bboxes.new_zeros((n, self.reg_dim))
        if num_pos > 0:
            labels[:num_pos] = pos_gt_labels.long()
            label_weights[:num_pos] = 1.0 if self.pos_weight <= 0 else self.pos_weight
            bbox_targets[:num_pos, :] = pos_gt_bboxes if self.reg_decoded_bbox else \
                self.bbox_coder.encode(pos_bboxes, pos_gt_bboxes)
            bbox_weights[:num_pos, :] = 1
        if num_neg > 0:
            label_weights[-num_neg:] = 1.0
        return labels, label_weights, bbox_targets, None, bbox_weights

    def get_bboxes_targets(self, sampling_results, concat=True, use_delta_and_decode=False):
        outs = multi_apply(self.get_bboxes_target_single, [r.pos_bboxes for r in sampling_results],
                           [r.neg_bboxes for r in sampling_results], [r.pos_gt_bboxes for r in sampling_results],
                           [r.pos_gt_labels for r in sampling_results], use_delta_and_decode=use_delta_and_decode)
        labels, lw, bt, _, bw = outs
        if concat:
            labels, lw, bt, bw = torch.cat(labels, 0), torch.cat(lw, 0), torch.cat(bt, 0), torch.cat(bw, 0)
        return labels, lw, bt, None, bw

    def get_bboxes(self, rois, cls_score, bbox_pred, img_shape, scale_factor, rescale=False):
        if isinstance(cls_score, list):
            cls_score = sum(cls_score) / float(len(cls_score))
        assert cls_score.dim() == 2, "Check cls_score.ndim"
        scores = F.softmax(cls_score, dim=-1)
        if bbox_pred is not None:
            bboxes = self.bbox_coder.decode(rois[:, 1:], bbox_pred, max_shape=img_shape)
        else:
            assert self.start_bbox_type == self.end_bbox_type
            bboxes = rois[:, 1:].clone()
        if rescale:
            sf = [scale_factor] * 4 if isinstance(scale_factor, float) else scale_factor
            sf = bboxes.new_tensor(sf)
            bboxes = bboxes.view(bboxes.size(0), -1, get_bbox_dim(self.end_bbox_type))
            if self.end_bbox_type == 'hbb':
                bboxes = bboxes / sf
            elif self.end_bbox_type == 'obb':
                bboxes = torch.cat([bboxes[..., :4] / sf, bboxes[..., 4:]], dim=-1)
            else:
                bboxes = bboxes / sf.repeat(2)
            bboxes = bboxes.view(bboxes.size(0), -1)
        return self.get_results(bboxes, scores, bbox_type=self.end_bbox_type)

    def _forward_train_masked(self, x, proposal_list, targets):
        """The training branch on FIXED-SIZE data, no host synchronisation: ``proposal_list[i]`` = (dets (P, 6), real
        (P,) bool) from OrientedRPNHead (its unused rows are neither positive nor negative), ``sampler.sample_masked``
        gives exactly ``num`` RoIs per image in the reference's order with masks for what the reference expresses by
        list lengths; targets and losses are the reference's sums over the same samples (:566-588, :426-496, :354-424).
        An unused slot still travels through RoIAlign and the FCs (as a copy of some real box) with all its weights 0."""
        dev = x[0].device
        fused = self._targets_fused(x, proposal_list, targets)
        if fused is not None:
            rois, labels, lweights, btargets, bweights, n_samples = fused
            scores, deltas, rois = self.forward_single(x, rois, test=None)
            return self.loss(scores, deltas, rois, labels, lweights, btargets, None, bweights, num_samples=n_samples)
        rois, labels, lweights, btargets, bweights, counts = [], [], [], [], [], []
        for i, t in enumerate(targets):
            obb = torch.as_tensor(t["rboxes"]).to(dev).float().clone()
            obb[:, -1] *= -1
            lab = torch.as_tensor(t["labels"]).to(dev) - 1
            props, real = proposal_list[i]
            ar = self.assigner.assign(props, obb, None, lab)
            ms = self.sampler.sample_masked(ar, props, obb, lab, valid=real)
            rois.append(torch.cat([ms.bboxes.new_full((ms.bboxes.size(0), 1), i), ms.bboxes], dim=-1))
            labels.append(torch.where(ms.is_pos, ms.pos_gt_labels.long(), torch.full_like(ms.inds, self.num_classes)))
            w = ms.valid.to(ms.bboxes.dtype)
            if self.pos_weight > 0:
                w = torch.where(ms.is_pos, torch.full_like(w, float(self.pos_weight)), w)
            lweights.append(w)
            bt = ms.pos_gt_bboxes if self.reg_decoded_bbox else self.bbox_coder.encode(ms.bboxes, ms.pos_gt_bboxes)
            btargets.append(torch.where(ms.is_pos[:, None], bt, torch.zeros_like(bt)))
            bweights.append(ms.is_pos[:, None].to(bt.dtype).expand(-1, self.reg_dim))
            counts.append(ms.n_pos + ms.n_neg)
        rois = torch.cat(rois, 0)
        scores, deltas, rois = self.forward_single(x, rois, test=None)
        return self.loss(scores, deltas, rois, torch.cat(labels), torch.cat(lweights), torch.cat(btargets), None,
                         torch.cat(bweights), num_samples=torch.stack(counts).sum())

    def _targets_fused(self, x, proposal_list, targets):
        """The loop of _forward_train_masked with the sampler as one radix select (6 launches) and everything between the
        samples and the RoI extractor -- gathers, arb2roi, labels, encoded targets, weights -- as one launch per image
        (csrc/orpn.hip), written straight into the batch's rows.  None when the configuration is not the one restated."""
        from rs_detection_amd.models.boxes.sampler import RandomSampler
        from rs_detection_amd.ops import orpn
        sampler, coder = self.sampler, self.bbox_coder
        N, num = len(targets), int(sampler.num)
        if not (orpn._ON and isinstance(sampler, RandomSampler) and sampler.add_gt_as_proposals and sampler.box_dim == 5
                and type(coder).__name__ == "OrientedDeltaXYWHTCoder" and not self.reg_decoded_bbox and self.reg_dim == 5
                and self.start_bbox_type == 'obb' and 0 < num <= 1024):
            return None
        dev = x[0].device
        rois = torch.empty((N * num, 6), dtype=torch.float32, device=dev)
        labels = torch.empty((N * num,), dtype=torch.int64, device=dev)
        lweights = torch.empty((N * num,), dtype=torch.float32, device=dev)
        btargets = torch.empty((N * num, 5), dtype=torch.float32, device=dev)
        bweights = torch.empty((N * num, 5), dtype=torch.float32, device=dev)
        inds = torch.empty((N, num), dtype=torch.int64, device=dev)
        assigned = torch.empty((N, num), dtype=torch.int64, device=dev)
        flags = torch.empty((2, N, num), dtype=torch.bool, device=dev)
        counts = torch.empty((N, 2), dtype=torch.int64, device=dev)
        for i, t in enumerate(targets):
            obb = torch.as_tensor(t["rboxes"]).to(dev).float().clone()
            obb[:, -1] *= -1
            lab = (torch.as_tensor(t["labels"]).to(dev) - 1).long()
            props, real = proposal_list[i]
            if not orpn.roi_targets_apply(props, obb, lab):
                return None
            ar = self.assigner.assign(props, obb, None, lab)
            K = obb.shape[0]
            pri = sampler.priorities(props.shape[0] + K, dev)
            if not orpn.sampler_applies(ar.gt_inds, pri, num):
                return None
            sample = (inds[i], flags[0, i], flags[1, i], assigned[i])
            orpn.sample_masked(ar.gt_inds, real, K, pri, num, int(sampler.num * sampler.pos_fraction), sampler.neg_pos_ub,
                               out=sample + (counts[i],))
            r = slice(i * num, (i + 1) * num)
            orpn.roi_targets(props, obb, lab, sample, i, self.num_classes, coder.means, coder.stds, self.pos_weight,
                             (rois[r], labels[r], lweights[r], btargets[r], bweights[r]))
        return rois, labels, lweights, btargets, bweights, counts.sum()

    def forward(self, x, proposal_list, targets):
        dev = x[0].device
        if self.training and len(proposal_list) and isinstance(proposal_list[0], tuple):
            return self._forward_train_masked(x, proposal_list, targets)
        if self.training:
            gt_obb, gt_labels = [], []
            for t in targets:
                obb = torch.as_tensor(t["rboxes"]).to(dev).float().clone()
                obb[:, -1] *= -1
                gt_obb.append(obb)
                gt_labels.append(torch.as_tensor(t["labels"]).to(dev) - 1)
            assert self.start_bbox_type == 'obb', "hbb start boxes are not on the Oriented-RCNN path"
            sampling_results = []
            for i in range(len(targets)):
                ar = self.assigner.assign(proposal_list[i], gt_obb[i], None, gt_labels[i])
                sampling_results.append(self.sampler.sample(ar, proposal_list[i], gt_obb[i], gt_labels[i]))
            scores, deltas, rois = self.forward_single(x, sampling_results, test=False)
            return self.loss(scores, deltas, rois, *self.get_bboxes_targets(sampling_results))
        result = []
        for i in range(len(targets)):
            # arb2roi numbers the RoIs of a one-element list as image 0: hand it image i's slice of the pyramid (the
            # reference passes the whole batch, oriented_head.py:610-613, which pools images i>0 from image 0)
            scores, deltas, rois = self.forward_single([f[i:i + 1] for f in x], [proposal_list[i]], test=True)
            det, labels = self.get_bboxes(rois, scores, deltas, targets[i]['img_size'], targets[i]['scale_factor'],
                                          rescale=True)
            result.append((det[:, :8], det[:, 8], labels))
        return result
