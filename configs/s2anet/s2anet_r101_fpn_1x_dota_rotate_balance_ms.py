# S2ANet-R101-FPN, multi-scale tiles, flip + ra90 rotate augmentation (BASELINE.json configs[4]).
# Same keys and values as the reference's projects/s2anet/configs/s2anet_r101_fpn_1x_dota_rotate_balance_ms.py
# (tests/test_configs_cpu.py checks the equality whenever the reference is mounted).  "ms" lives in the OFFLINE
# preprocessing: the images are rescaled by 0.5 / 1.0 / 1.5 and cut into 1024x1024 tiles with 200 px overlap
# (dataset_dir .../preprocessed_ms/train_1024_200_0.5-1.0-1.5), so the network always sees 1024^2 tiles.
_base_ = None
_focal = dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0)
_smooth_l1 = dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0)
_norm = dict(type="Normalize", mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_bgr=False)
_resize = dict(type="RotatedResize", min_size=1024, max_size=1024)
_pad = dict(type="Pad", size_divisor=32)


def _stage_cfg():
    return dict(
        assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1,
                      iou_calculator=dict(type='BboxOverlaps2D_rotated')),
        bbox_coder=dict(type='DeltaXYWHABBoxCoder', target_means=(0., 0., 0., 0., 0.),
                        target_stds=(1., 1., 1., 1., 1.), clip_border=True),
        allowed_border=-1, pos_weight=-1, debug=False)


model = dict(
    type='S2ANet',
    backbone=dict(type='Resnet101', frozen_stages=1, return_stages=["layer1", "layer2", "layer3", "layer4"],
                  pretrained=True),
    neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
              add_extra_convs="on_input", num_outs=5),
    bbox_head=dict(
        type='S2ANetHead', num_classes=16, in_channels=256, feat_channels=256, stacked_convs=2, with_orconv=True,
        anchor_ratios=[1.0], anchor_strides=[8, 16, 32, 64, 128], anchor_scales=[4],
        target_means=[.0, .0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0, 1.0],
        loss_fam_cls=dict(_focal), loss_fam_bbox=dict(_smooth_l1),
        loss_odm_cls=dict(_focal), loss_odm_bbox=dict(_smooth_l1),
        test_cfg=dict(nms_pre=2000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms_rotated', iou_thr=0.1),
                      max_per_img=2000),
        train_cfg=dict(fam_cfg=_stage_cfg(), odm_cfg=_stage_cfg())))

dataset_root = '/media/data3/lyx/Detection'
dataset = dict(
    train=dict(type="FAIR1M_1_5_Dataset",
               dataset_dir=f'{dataset_root}/preprocessed_ms/train_1024_200_0.5-1.0-1.5',
               transforms=[dict(_resize), dict(type='RotatedRandomFlip', prob=0.5),
                           dict(type="RandomRotateAug", random_rotate_on=True), dict(_pad), dict(_norm)],
               batch_size=8, num_workers=8, shuffle=True, filter_empty_gt=False),
    val=dict(type="FAIR1M_1_5_Dataset",
             dataset_dir=f'{dataset_root}/preprocessed_ms/train_1024_200_0.5-1.0-1.5',
             transforms=[dict(_resize), dict(_pad), dict(_norm)], batch_size=8, num_workers=8, shuffle=False),
    test=dict(type="ImageDataset", images_dir=f'{dataset_root}/preprocessed_ms/test_1024_200_0.5-1.0-1.5/images',
              transforms=[dict(_resize), dict(_pad), dict(_norm)], dataset_type="FAIR1M_1_5", num_workers=4,
              batch_size=1))

optimizer = dict(type='SGD', lr=0.01 / 4., momentum=0.9, weight_decay=0.0001,
                 grad_clip=dict(max_norm=35, norm_type=2))
scheduler = dict(type='StepLR', warmup='linear', warmup_iters=500, warmup_ratio=1.0 / 3, milestones=[7, 10])
logger = dict(type="RunLogger")

max_epoch = 12
eval_interval = 3
checkpoint_interval = 1
log_interval = 50

del _focal, _smooth_l1, _stage_cfg, _norm, _resize, _pad, _base_
