# RetinaNet-hbb R50-FPN (BASELINE config[0]: 2 x 600 x 600 tiles, horizontal boxes, no rotated op, CPU-runnable).
# Model section = the projects/retinanet config of the reference with mode 'H' and horizontal anchors.
model = dict(
    type="RetinaNet",
    backbone=dict(
        type='Resnet50',
        frozen_stages=1,
        return_stages=["layer1", "layer2", "layer3", "layer4"],
        pretrained=False),
    neck=dict(
        type="FPN",
        in_channels=[256, 512, 1024, 2048],
        out_channels=256,
        start_level=1,
        add_extra_convs="on_input",
        num_outs=5),
    rpn_net=dict(
        type="RetinaHead",
        n_class=15,
        in_channels=256,
        stacked_convs=4,
        mode="H",
        score_threshold=0.05,
        nms_iou_threshold=0.5,
        max_dets=100,
        roi_beta=1 / 9.,
        cls_loss_weight=1.,
        loc_loss_weight=0.2,
        anchor_generator=dict(
            type="AnchorGeneratorRotated",
            strides=[8, 16, 32, 64, 128],
            ratios=[0.5, 1.0, 2.0],
            scales=[4., 5.0396842, 6.34960421],
            mode="H")),
)
optimizer = dict(type='SGD', lr=0.005, momentum=0.9, weight_decay=0.0001)
scheduler = dict(type='StepLR', warmup='linear', warmup_iters=500, warmup_ratio=0.001, milestones=[7, 10])
max_epoch = 12
