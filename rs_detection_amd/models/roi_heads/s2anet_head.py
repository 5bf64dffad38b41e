"""S2ANetHead on MI355X (/root/reference/python/jdet/models/roi_heads/s2anet_head.py:20-723).

Same constructor arguments, parameter names, init, losses and outputs as the
reference; the per-image / per-gt Python loops are replaced by batched device work:

  forward_single  fam convs (MIOpen) -> [bbox_decode + AlignConv.get_offset] fused in ONE
                  HIP kernel for the whole batch (:631-654 + :676-713) -> DeformConv
                  (csrc/deform_conv.hip + rocBLAS GEMM) -> ORConv2d (csrc/arf.hip + MIOpen)
                  -> RotationInvariantPooling -> odm convs.
  loss            FAM and ODM targets each by ONE grouped rotated-IoU launch + ONE
                  assignment call for all images (anchor_target_batched); losses on the
                  level-concatenated tensors; no host synchronisation anywhere.
"""
import numpy as np
import torch
import torch.nn as nn

from rs_detection_amd.models.boxes.anchor_generator import AnchorGeneratorRotatedS2ANet
from rs_detection_amd.models.boxes.anchor_target import anchor_target_batched
from rs_detection_amd.models.utils.modules import ConvModule
from rs_detection_amd.models.utils.weight_init import normal_init, bias_init_with_prob
from rs_detection_amd.ops import (DeformConv, ORConv2d, RotationInvariantPooling, multiclass_nms_rotated,
                                  delta2bbox_rotated, rotated_box_to_poly, s2a_refine_and_offset)
from rs_detection_amd.utils.registry import HEADS, LOSSES, BOXES, build_from_cfg

_DEFAULT_TRAIN_PART = dict(
    assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1,
                  iou_calculator=dict(type='BboxOverlaps2D_rotated')),
    bbox_coder=dict(type='DeltaXYWHABBoxCoder', target_means=(0., 0., 0., 0., 0.),
                    target_stds=(1., 1., 1., 1., 1.), clip_border=True),
    allowed_border=-1, pos_weight=-1, debug=False)


def bbox_decode(bbox_preds, anchors, means=(0, 0, 0, 0, 0), stds=(1, 1, 1, 1, 1)):
    """s2anet_head.py:631-654: (N,5,H,W) deltas + (H*W,5) anchors -> (N,H,W,5), wh_ratio_clip=1e-6."""
    refined, _ = s2a_refine_and_offset(bbox_preds, anchors, 1.0, 3, means, stds, 1e-6, want_offset=False)
    return refined


class AlignConv(nn.Module):
    """s2anet_head.py:657-723."""

    def __init__(self, in_channels, out_channels, kernel_size=3, deformable_groups=1):
        super().__init__()
        self.kernel_size = kernel_size
        self.deform_conv = DeformConv(in_channels, out_channels, kernel_size=kernel_size,
                                      padding=(kernel_size - 1) // 2, deformable_groups=deformable_groups)
        self.relu = nn.ReLU(inplace=True)

    def init_weights(self):
        normal_init(self.deform_conv, std=0.01)

    @torch.no_grad()
    def get_offset(self, anchors, featmap_size, stride):
        """Reference signature (:676): one image's (H*W,5) anchors -> (2*ks*ks, H, W).
        (Pure torch, used by tests/tools; the training path uses the fused kernel.)"""
        feat_h, feat_w = featmap_size
        ks = self.kernel_size
        pad = (ks - 1) // 2
        idx = torch.arange(-pad, pad + 1, dtype=anchors.dtype, device=anchors.device)
        yy, xx = torch.meshgrid(idx, idx, indexing="ij")
        xx, yy = xx.reshape(-1), yy.reshape(-1)
        yc, xc = torch.meshgrid(torch.arange(feat_h, dtype=anchors.dtype, device=anchors.device),
                                torch.arange(feat_w, dtype=anchors.dtype, device=anchors.device), indexing="ij")
        x_conv, y_conv = xc.reshape(-1)[:, None] + xx, yc.reshape(-1)[:, None] + yy
        x_ctr, y_ctr, w, h, a = torch.unbind(anchors, dim=1)
        x_ctr, y_ctr, w, h = x_ctr / stride, y_ctr / stride, w / stride, h / stride
        cos, sin = torch.cos(a), torch.sin(a)
        x, y = (w / ks)[:, None] * xx, (h / ks)[:, None] * yy
        xr = cos[:, None] * x - sin[:, None] * y
        yr = sin[:, None] * x + cos[:, None] * y
        off_x = xr + x_ctr[:, None] - x_conv
        off_y = yr + y_ctr[:, None] - y_conv
        off = torch.stack([off_y, off_x], dim=-1)
        return off.reshape(anchors.size(0), -1).permute(1, 0).reshape(-1, feat_h, feat_w)

    def forward(self, x, anchors, stride, offset=None):
        if offset is None:
            num_imgs, H, W = anchors.shape[:3]
            offset = torch.stack([self.get_offset(anchors[i].reshape(-1, 5), (H, W), stride)
                                  for i in range(num_imgs)], dim=0)
        return self.relu(self.deform_conv(x, offset))


@HEADS.register_module()
class S2ANetHead(nn.Module):
    def __init__(self, num_classes, in_channels, feat_channels=256, stacked_convs=2, with_orconv=True,
                 anchor_scales=[4], anchor_ratios=[1.0], anchor_strides=[8, 16, 32, 64, 128], anchor_base_sizes=None,
                 target_means=(.0, .0, .0, .0, .0), target_stds=(1.0, 1.0, 1.0, 1.0, 1.0),
                 loss_fam_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_fam_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
                 loss_odm_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_odm_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
                 test_cfg=dict(nms_pre=2000, min_bbox_size=0, score_thr=0.05,
                               nms=dict(type='nms_rotated', iou_thr=0.1), max_per_img=2000),
                 train_cfg=dict(fam_cfg=_DEFAULT_TRAIN_PART, odm_cfg=_DEFAULT_TRAIN_PART)):
        super().__init__()
        self.num_classes, self.in_channels, self.feat_channels = num_classes, in_channels, feat_channels
        self.stacked_convs, self.with_orconv = stacked_convs, with_orconv
        self.anchor_scales, self.anchor_ratios, self.anchor_strides = anchor_scales, anchor_ratios, list(anchor_strides)
        self.anchor_base_sizes = list(anchor_strides) if anchor_base_sizes is None else anchor_base_sizes
        self.target_means, self.target_stds = tuple(target_means), tuple(target_stds)
        self.use_sigmoid_cls = loss_odm_cls.get('use_sigmoid', False)
        self.sampling = loss_odm_cls['type'] not in ['FocalLoss', 'GHMC']
        self.cls_out_channels = num_classes - 1 if self.use_sigmoid_cls else num_classes
        if self.cls_out_channels <= 0:
            raise ValueError('num_classes={} is too small'.format(num_classes))
        self.loss_fam_cls = build_from_cfg(loss_fam_cls, LOSSES)
        self.loss_fam_bbox = build_from_cfg(loss_fam_bbox, LOSSES)
        self.loss_odm_cls = build_from_cfg(loss_odm_cls, LOSSES)
        self.loss_odm_bbox = build_from_cfg(loss_odm_bbox, LOSSES)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.anchor_generators = [AnchorGeneratorRotatedS2ANet(b, anchor_scales, anchor_ratios)
                                  for b in self.anchor_base_sizes]
        self.base_anchors = dict()  # anchor cache, keyed (level, featmap_size, device)
        self.packed = True          # shared-weight towers on the pyramid canvas (forward_packed); False = the level loop
        self.canvas_groups = None   # None = by dtype (_level_groups); 'all' | 'split'
        self._built = {}
        self._init_anchor_cat = {}   # level-concatenated grid per (featmap sizes, device)
        self._init_layers()

    def _init_layers(self):
        fc = self.feat_channels
        self.relu = nn.ReLU(inplace=True)
        self.fam_reg_convs, self.fam_cls_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(self.stacked_convs):
            chn = self.in_channels if i == 0 else fc
            self.fam_reg_convs.append(ConvModule(chn, fc, 3, stride=1, padding=1))
            self.fam_cls_convs.append(ConvModule(chn, fc, 3, stride=1, padding=1))
        self.fam_reg = nn.Conv2d(fc, 5, 1)
        self.fam_cls = nn.Conv2d(fc, self.cls_out_channels, 1)
        self.align_conv = AlignConv(fc, fc, kernel_size=3)
        if self.with_orconv:
            self.or_conv = ORConv2d(fc, int(fc / 8), kernel_size=3, padding=1, arf_config=(1, 8))
        else:
            self.or_conv = nn.Conv2d(fc, fc, 3, padding=1)
        self.or_pool = RotationInvariantPooling(256, 8)
        self.odm_reg_convs, self.odm_cls_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(self.stacked_convs):
            chn = int(fc / 8) if i == 0 and self.with_orconv else fc
            self.odm_reg_convs.append(ConvModule(fc, fc, 3, stride=1, padding=1))
            self.odm_cls_convs.append(ConvModule(chn, fc, 3, stride=1, padding=1))
        self.odm_cls = nn.Conv2d(fc, self.cls_out_channels, 3, padding=1)
        self.odm_reg = nn.Conv2d(fc, 5, 3, padding=1)
        self.init_weights()

    def init_weights(self):
        for m in list(self.fam_reg_convs) + list(self.fam_cls_convs):
            normal_init(m.conv, std=0.01)
        bias_cls = bias_init_with_prob(0.01)
        normal_init(self.fam_reg, std=0.01)
        normal_init(self.fam_cls, std=0.01, bias=bias_cls)
        self.align_conv.init_weights()
        normal_init(self.or_conv, std=0.01)
        for m in list(self.odm_reg_convs) + list(self.odm_cls_convs):
            normal_init(m.conv, std=0.01)
        normal_init(self.odm_cls, std=0.01, bias=bias_cls)
        normal_init(self.odm_reg, std=0.01)

    def _anchors(self, level, featmap_size, device):
        key = (level, tuple(featmap_size), str(device))
        if key not in self.base_anchors:
            self.base_anchors[key] = self.anchor_generators[level].grid_anchors(
                featmap_size, self.anchor_strides[level], device=device)
        return self.base_anchors[key]

    def forward_single(self, x, stride):
        fam_reg_feat = x
        for conv in self.fam_reg_convs:
            fam_reg_feat = conv(fam_reg_feat)
        fam_bbox_pred = self.fam_reg(fam_reg_feat)
        if self.training:  # only forward during training (:214-218)
            fam_cls_feat = x
            for conv in self.fam_cls_convs:
                fam_cls_feat = conv(fam_cls_feat)
            fam_cls_score = self.fam_cls(fam_cls_feat)
        else:
            fam_cls_score = None
        level = self.anchor_strides.index(stride)
        featmap_size = tuple(fam_bbox_pred.shape[-2:])
        init_anchors = self._anchors(level, featmap_size, x.device)
        # bbox_decode(fam_bbox_pred.detach(), ...) + AlignConv.get_offset, fused, whole batch
        refine_anchor, offset = s2a_refine_and_offset(fam_bbox_pred.detach().float(), init_anchors, stride, 3,
                                                      self.target_means, self.target_stds, 1e-6)
        align_feat = self.align_conv(x, refine_anchor, stride, offset=offset)
        or_feat = self.or_conv(align_feat)
        odm_reg_feat = or_feat
        odm_cls_feat = self.or_pool(or_feat) if self.with_orconv else or_feat
        for conv in self.odm_reg_convs:
            odm_reg_feat = conv(odm_reg_feat)
        for conv in self.odm_cls_convs:
            odm_cls_feat = conv(odm_cls_feat)
        return fam_cls_score, fam_bbox_pred, refine_anchor, self.odm_cls(odm_cls_feat), self.odm_reg(odm_reg_feat)

    # ---- targets -------------------------------------------------------------------
    def _valid_flags(self, featmap_sizes, img_metas, device):
        """None when every anchor of every image is valid (the common, padded-to-stride case)."""
        need = False
        per_img = []
        for meta in img_metas:
            w, h = meta['pad_shape'][:2]  # (w,h) convention kept
            flags = []
            for i, (fh, fw) in enumerate(featmap_sizes):
                s = self.anchor_strides[i]
                vh, vw = min(int(np.ceil(h / s)), fh), min(int(np.ceil(w / s)), fw)
                need |= (vh < fh) or (vw < fw)
                flags.append((fh, fw, vh, vw))
            per_img.append(flags)
        if not need:
            return None
        out = []
        for flags in per_img:
            out.append(torch.cat([self.anchor_generators[i].valid_flags((fh, fw), (vh, vw), device)
                                  for i, (fh, fw, vh, vw) in enumerate(flags)]))
        return torch.stack(out)

    def _cfg_objs(self, name):
        if name not in self._built:
            cfg = self.train_cfg[name]
            coder_cfg = cfg.get('bbox_coder', '')
            coder = build_from_cfg(dict(type='DeltaXYWHABBoxCoder') if coder_cfg == '' else coder_cfg, BOXES)
            self._built[name] = (build_from_cfg(cfg.get('assigner', ''), BOXES), coder)
        return self._built[name]

    @staticmethod
    def _flatten(preds, ch):
        """list of (B,ch,H,W) per level -> (B, A, ch) level-concatenated, anchor order = (h, w)."""
        return torch.cat([p.permute(0, 2, 3, 1).reshape(p.shape[0], -1, ch) for p in preds], dim=1)

    def _fused_losses(self, cls_loss, bbox_loss, cls_maps, box_maps, labels, label_weights, bbox_targets, bbox_weights,
                      avg, cfg):
        """Both losses of one module over all levels as one HIP pass each way (ops/s2a_loss.py, csrc/losses.hip) --
        when the configured modules are the sigmoid FocalLoss + SmoothL1Loss with 'mean' reduction, the maps are on the
        GPU in one storage type, and the regression loss is on the encoded deltas.  None otherwise."""
        from rs_detection_amd.models.losses.focal_loss import FocalLoss
        from rs_detection_amd.models.losses.smooth_l1_loss import SmoothL1Loss
        if type(cls_loss) is not FocalLoss or type(bbox_loss) is not SmoothL1Loss:
            return None
        if cls_loss.reduction != 'mean' or bbox_loss.reduction != 'mean' or cfg.get('reg_decoded_bbox', False):
            return None
        dt = cls_maps[0].dtype
        if not cls_maps[0].is_cuda or dt not in (torch.float32, torch.bfloat16) or \
                any(m.dtype != dt for m in list(cls_maps) + list(box_maps)):
            return None
        from rs_detection_amd.ops.s2a_loss import s2a_level_losses
        out = s2a_level_losses(cls_maps, box_maps, labels, label_weights, bbox_targets, bbox_weights, avg,
                               cls_loss.alpha, cls_loss.gamma, bbox_loss.beta, cls_loss.loss_weight,
                               bbox_loss.loss_weight)
        return list(out[0].unbind(0)), list(out[1].unbind(0))

    def _level_losses(self, cls_loss, bbox_loss, cls_score, bbox_pred, labels, label_weights, bbox_targets,
                      bbox_weights, num_level_anchors, avg):
        """Per-level loss lists like the reference's multi_apply(loss_*_single) (:430-508)."""
        l_cls, l_box, s = [], [], 0
        C = self.cls_out_channels
        for n in num_level_anchors:
            sl = slice(s, s + n)
            l_cls.append(cls_loss(cls_score[:, sl].reshape(-1, C), labels[:, sl].reshape(-1),
                                  label_weights[:, sl].reshape(-1), avg_factor=avg))
            l_box.append(bbox_loss(bbox_pred[:, sl].reshape(-1, 5), bbox_targets[:, sl].reshape(-1, 5),
                                   bbox_weights[:, sl].reshape(-1, 5), avg_factor=avg))
            s += n
        return l_cls, l_box

    def loss(self, fam_cls_scores, fam_bbox_preds, refine_anchors, odm_cls_scores, odm_bbox_preds, gt_bboxes,
             gt_labels, img_metas, gt_bboxes_ignore=None):
        device = odm_cls_scores[0].device
        featmap_sizes = [tuple(f.shape[-2:]) for f in odm_cls_scores]
        assert len(featmap_sizes) == len(self.anchor_generators)
        num_level_anchors = [h * w for h, w in featmap_sizes]
        ks = [int(g.shape[0]) for g in gt_bboxes]
        if min(ks) == 0:
            raise ValueError('No gt or bboxes')  # assigner.py:91-92
        gt_cat = torch.cat([g.to(device=device, dtype=torch.float32) for g in gt_bboxes])
        lab_cat = torch.cat([l.to(device=device, dtype=torch.int32) for l in gt_labels])
        row_offsets = torch.tensor(np.concatenate([[0], np.cumsum(ks)]), dtype=torch.int32).to(device, non_blocking=True)
        valid = self._valid_flags(featmap_sizes, img_metas, device)
        C = self.cls_out_channels

        # Feature Alignment Module: shared grid anchors (one tensor per pyramid geometry, kept: its prepared form --
        # fp64 sincos, strip boxes -- is then computed once per process, ops/anchor_target.prepare_boxes)
        akey = (tuple(featmap_sizes), str(device))
        if akey not in self._init_anchor_cat:
            self._init_anchor_cat[akey] = torch.cat([self._anchors(i, featmap_sizes[i], device)
                                                     for i in range(len(featmap_sizes))])
        init_anchors = self._init_anchor_cat[akey]
        assigner, coder = self._cfg_objs('fam_cfg')
        from rs_detection_amd.ops.anchor_target import prepare_boxes
        # the gts' prepared form (fp64 sincos) once for both modules; not cached (these gts never come back)
        pgt = prepare_boxes(gt_cat, heavy_from=gt_cat.shape[0]) if gt_cat.is_cuda else None
        labels, lw, bt, bw, npos, nneg = anchor_target_batched(init_anchors, gt_cat, lab_cat, row_offsets, max(ks),
                                                               self.train_cfg['fam_cfg'], assigner, coder, valid,
                                                               ks=ks, cache_anchors=True, prepared_gt=pgt)
        avg = npos + nneg if self.sampling else npos
        fused = self._fused_losses(self.loss_fam_cls, self.loss_fam_bbox, fam_cls_scores, fam_bbox_preds, labels, lw, bt,
                                   bw, avg, self.train_cfg['fam_cfg'])
        losses_fam_cls, losses_fam_bbox = fused if fused is not None else self._level_losses(
            self.loss_fam_cls, self.loss_fam_bbox, self._flatten(fam_cls_scores, C), self._flatten(fam_bbox_preds, 5),
            labels, lw, bt, bw, num_level_anchors, avg)

        # Oriented Detection Module: per-image refined anchors
        refined = torch.cat([r.reshape(r.shape[0], -1, 5) for r in refine_anchors], dim=1)
        assigner, coder = self._cfg_objs('odm_cfg')
        heavy = prepare_boxes(init_anchors, cache=True).heavy_from     # the refinements keep the grid's level layout
        labels, lw, bt, bw, npos, nneg = anchor_target_batched(refined, gt_cat, lab_cat, row_offsets, max(ks),
                                                               self.train_cfg['odm_cfg'], assigner, coder, valid,
                                                               ks=ks, heavy_from=heavy, prepared_gt=pgt)
        avg = npos + nneg if self.sampling else npos
        fused = self._fused_losses(self.loss_odm_cls, self.loss_odm_bbox, odm_cls_scores, odm_bbox_preds, labels, lw, bt,
                                   bw, avg, self.train_cfg['odm_cfg'])
        losses_odm_cls, losses_odm_bbox = fused if fused is not None else self._level_losses(
            self.loss_odm_cls, self.loss_odm_bbox, self._flatten(odm_cls_scores, C), self._flatten(odm_bbox_preds, 5),
            labels, lw, bt, bw, num_level_anchors, avg)
        return dict(loss_fam_cls=losses_fam_cls, loss_fam_bbox=losses_fam_bbox, loss_odm_cls=losses_odm_cls,
                    loss_odm_bbox=losses_odm_bbox)

    # ---- inference -----------------------------------------------------------------
    def get_bboxes(self, fam_cls_scores, fam_bbox_preds, refine_anchors, odm_cls_scores, odm_bbox_preds, img_metas,
                   rescale=True):
        assert len(odm_cls_scores) == len(odm_bbox_preds)
        cfg = self.test_cfg
        num_levels = len(odm_cls_scores)
        results = []
        for img_id in range(len(img_metas)):
            cls_list = [odm_cls_scores[i][img_id].detach() for i in range(num_levels)]
            box_list = [odm_bbox_preds[i][img_id].detach() for i in range(num_levels)]
            anchors = [refine_anchors[i][img_id].reshape(-1, 5) for i in range(num_levels)]
            results.append(self.get_bboxes_single(cls_list, box_list, anchors, img_metas[img_id]['img_shape'],
                                                  img_metas[img_id]['scale_factor'], cfg, rescale))
        return results

    def get_bboxes_single(self, cls_score_list, bbox_pred_list, mlvl_anchors, img_shape, scale_factor, cfg,
                          rescale=False):
        assert len(cls_score_list) == len(bbox_pred_list) == len(mlvl_anchors)
        mlvl_bboxes, mlvl_scores = [], []
        for cls_score, bbox_pred, anchors in zip(cls_score_list, bbox_pred_list, mlvl_anchors):
            assert cls_score.shape[-2:] == bbox_pred.shape[-2:]
            cls_score = cls_score.permute(1, 2, 0).reshape(-1, self.cls_out_channels)
            scores = cls_score.sigmoid() if self.use_sigmoid_cls else cls_score.softmax(-1)
            bbox_pred = bbox_pred.permute(1, 2, 0).reshape(-1, 5)
            nms_pre = cfg.get('nms_pre', -1)
            if nms_pre > 0 and scores.shape[0] > nms_pre:
                max_scores = scores.max(dim=1)[0] if self.use_sigmoid_cls else scores[:, 1:].max(dim=1)[0]
                _, topk = max_scores.topk(nms_pre)
                anchors, bbox_pred, scores = anchors[topk, :], bbox_pred[topk, :], scores[topk, :]
            mlvl_bboxes.append(delta2bbox_rotated(anchors, bbox_pred.float(), self.target_means, self.target_stds,
                                                  img_shape))
            mlvl_scores.append(scores)
        mlvl_bboxes = torch.cat(mlvl_bboxes)
        if rescale:
            mlvl_bboxes[..., :4] /= scale_factor
        mlvl_scores = torch.cat(mlvl_scores)
        if self.use_sigmoid_cls:
            mlvl_scores = torch.cat([mlvl_scores.new_zeros(mlvl_scores.shape[0], 1), mlvl_scores], dim=1)
        det_bboxes, det_labels = multiclass_nms_rotated(mlvl_bboxes, mlvl_scores.float(), cfg['score_thr'], cfg['nms'],
                                                        cfg['max_per_img'])
        boxes, scores = det_bboxes[:, :5], det_bboxes[:, 5]
        return rotated_box_to_poly(boxes.contiguous()), scores, det_labels

    def parse_targets(self, targets, is_train=True):
        img_metas, gt_bboxes, gt_bboxes_ignore, gt_labels = [], [], [], []
        for t in targets:
            if is_train:
                gt_bboxes.append(torch.as_tensor(t["rboxes"]))
                gt_labels.append(torch.as_tensor(t["labels"]))
                gt_bboxes_ignore.append(t.get("rboxes_ignore"))
            img_metas.append(dict(img_shape=tuple(t["img_size"])[::-1], scale_factor=t["scale_factor"],
                                  pad_shape=t["pad_shape"]))
        if not is_train:
            return img_metas
        return gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore

    def _packed_ok(self, feats):
        """The canvas path (forward_packed) applies to GPU feature maps of one dtype / shape family; CPU tensors and a
        head with ``packed = False`` (tests, the bench's flop count) take the reference's per-level loop."""
        if not self.packed or len(feats) < 2 or len(feats) > 8:
            return False
        f0 = feats[0]
        return (f0.is_cuda and f0.dtype in (torch.float32, torch.bfloat16)
                and all(f.dim() == 4 and f.dtype == f0.dtype and f.shape[:2] == f0.shape[:2] for f in feats)
                and all(c.conv.kernel_size == (3, 3) and c.conv.padding == (1, 1) and c.conv.stride == (1, 1)
                        and c.conv.dilation == (1, 1)
                        for c in list(self.fam_reg_convs) + list(self.fam_cls_convs) + list(self.odm_reg_convs)
                        + list(self.odm_cls_convs)))

    @staticmethod
    def _tower(convs, x, lay):
        """The stacked ConvModules of one tower on the canvas (:207-252 of the reference runs them level by level).  A
        pair of plain conv + bias + ReLU layers that our 3x3 kernel takes runs as ONE autograd node whose backward folds
        the first layer's ReLU gate into the second layer's backward-data (ops/conv3x3.py: _Conv3x3Tower2)."""
        from rs_detection_amd.ops.conv3x3 import conv3x3_tower2, conv3x3_tower2_applies
        if (len(convs) == 2 and torch.is_grad_enabled() and all(m._fused_bias_relu(x, True) for m in convs)
                and conv3x3_tower2_applies(x, convs[0].conv, convs[1].conv)):
            return conv3x3_tower2(x, convs[0].conv, convs[1].conv, lay.live)
        for conv in convs:
            x = conv(x, canvas=lay)
        return x

    def forward_packed(self, feats, first_level=0):
        """forward_single (:207-252) for SEVERAL levels at once: the level maps laid side by side in one canvas
        (ops/pyramid.py), every shared-weight convolution of the FAM and ODM towers run once on it -- 13 convolution
        calls instead of 13 per level, and no 8 x 8 .. 32 x 32 launches that cannot fill the GPU.  The zero gap between
        neighbouring levels is each level's zero padding; the tower epilogues put it back after every convolution.
        AlignConv samples by per-level offsets that may reach far outside its own map (they must read zeros there, not a
        neighbour on the canvas), so it stays per level and reads the FPN maps themselves.  Same outputs as the loop:
        five lists of per-level maps.  ``feats`` are the maps of levels first_level, first_level + 1, ..."""
        from rs_detection_amd.ops.pyramid import canvas_layout, pyramid_pack, pyramid_unpack, canvas_bias_act
        from rs_detection_amd.ops.bn_act import conv2d_bias
        import torch.nn.functional as F
        sizes = [tuple(f.shape[-2:]) for f in feats]
        strides = self.anchor_strides[first_level:first_level + len(feats)]
        lay = canvas_layout(sizes, feats[0].device)
        xc = pyramid_pack(feats, lay)
        reg = self._tower(self.fam_reg_convs, xc, lay)
        fam_bbox_preds = pyramid_unpack(conv2d_bias(self.fam_reg, reg), lay)
        if self.training:
            cls = self._tower(self.fam_cls_convs, xc, lay)
            fam_cls_scores = pyramid_unpack(conv2d_bias(self.fam_cls, cls), lay)
        else:
            fam_cls_scores = [None] * len(feats)
        # bbox_decode + AlignConv.get_offset of every level in one launch, the bf16 predictions of an autocast step read as
        # they are (per level: a widening cast + a 5 us launch each)
        from rs_detection_amd.ops.box_coder import s2a_refine_and_offset_levels
        init_anchors = [self._anchors(first_level + i, sizes[i], feats[i].device) for i in range(len(feats))]
        refine_anchors, offsets = s2a_refine_and_offset_levels([p.detach() for p in fam_bbox_preds], init_anchors, strides, 3,
                                                               self.target_means, self.target_stds, 1e-6)
        align = [self.align_conv(x, ra, stride, offset=off)
                 for x, ra, stride, off in zip(feats, refine_anchors, strides, offsets)]
        ac = pyramid_pack(align, lay, channels_last=not xc.is_contiguous())
        oc = self.or_conv
        w = oc.rotate_arf() if isinstance(oc, ORConv2d) else oc.weight
        from rs_detection_amd.ops.conv3x3 import conv3x3_applies, conv3x3_same
        if conv3x3_applies(ac, w, oc.stride, oc.padding, oc.dilation, oc.groups):
            or_feat = conv3x3_same(ac, w)
        else:
            or_feat = F.conv2d(ac, w, None, oc.stride, oc.padding, oc.dilation, oc.groups)
        if oc.bias is not None:
            or_feat = canvas_bias_act(or_feat, oc.bias, lay, relu=False)
        else:
            or_feat = or_feat * lay.live_f.to(or_feat.dtype)
        odm_reg_feat = or_feat
        odm_cls_feat = self.or_pool(or_feat) if self.with_orconv else or_feat
        odm_reg_feat = self._tower(self.odm_reg_convs, odm_reg_feat, lay)
        odm_cls_feat = self._tower(self.odm_cls_convs, odm_cls_feat, lay)
        odm_cls_scores = pyramid_unpack(conv2d_bias(self.odm_cls, odm_cls_feat), lay)
        odm_bbox_preds = pyramid_unpack(conv2d_bias(self.odm_reg, odm_reg_feat), lay)
        return fam_cls_scores, fam_bbox_preds, refine_anchors, odm_cls_scores, odm_bbox_preds

    def _level_groups(self, n, dtype=torch.float32):
        """How the n levels are grouped into canvases.  'all' = one canvas; 'split' = level 0 on its own (through
        forward_single: no canvas, no gap pixels to pay for) + one canvas for the small levels.  Measured on the step
        (4 x 1024^2, profiles/r03_canvas_ab.txt): the fp32 step is GPU-bound and the all-levels canvas has 15 % more
        pixels than the five maps together, so 'split' wins there (57.3 -> 55.9 ms; 'all' 59.6); the bf16 step is bound
        by launches, so 'all' wins (22.7 -> 19.9 ms; 'split' 20.3).  ``self.canvas_groups`` ('all' | 'split') overrides."""
        mode = self.canvas_groups or ("all" if dtype == torch.bfloat16 else "split")
        if mode == "split" and n > 2:
            return [(0, 1), (1, n)]
        return [(0, n)]

    def forward_levels(self, feats):
        outs = [[], [], [], [], []]
        packed = self._packed_ok(feats) and len(feats) == len(self.anchor_strides)
        groups = self._level_groups(len(feats), feats[0].dtype) if packed else [(i, i + 1) for i in range(len(feats))]
        from rs_detection_amd.ops.dcn_v1 import shared_gather_index
        with shared_gather_index():     # the AlignConv backwards of all levels share one col2im index build
            for a, b in groups:
                if b - a == 1:
                    o = [[v] for v in self.forward_single(feats[a], self.anchor_strides[a])]
                else:
                    o = self.forward_packed(feats[a:b], first_level=a)
                for dst, src in zip(outs, o):
                    dst.extend(src)
        return tuple(outs)

    def forward(self, feats, targets):
        outs = self.forward_levels(list(feats)[:len(self.anchor_strides)])
        if self.training:
            return self.loss(*outs, *self.parse_targets(targets))
        return self.get_bboxes(*outs, self.parse_targets(targets, is_train=False))
