"""Does the assembled model LEARN?  Train on the rendered synthetic DOTA-format set (data/synthetic.py, render=True) and
report the loss curve + mAP on the training images.  python profiles/scripts/learn_proof.py [s2anet|orcnn] [f32|bf16] [iters]
Environment: TILE (256; 1024 = the bench's tile, whose 196-wide head canvas takes the conv3x3_mfma kernels), MF=channels_last
(the bench's bf16 layout), BACKBONE, LR."""
import os
import sys
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rs_detection_amd.config import Config  # noqa: E402
from rs_detection_amd.runner.runner import Runner  # noqa: E402


def make_runner(model="s2anet", dtype="f32", tile=256, classes=4, lr=None, images=32, batch=4, backbone=None):
    warnings.simplefilter("ignore", RuntimeWarning)
    if model == "s2anet":
        cfg = Config(os.path.join(ROOT, "configs", "s2anet", "s2anet_r50_fpn_1x_dota.py"))
        # from scratch: BN must train -- unless NORM_EVAL=1, which keeps the shipped configs' eval-mode BatchNorm (running
        # statistics at their initial 0 / 1: the BatchNorms are per-channel affine layers) so that the bf16 trunk takes the
        # round-5 routes that exist for eval-mode BatchNorm only (fused 1x1 conv + BN, the one-node Bottleneck)
        ne = os.environ.get("NORM_EVAL", "0") == "1"
        cfg.model["backbone"].update(pretrained=False, frozen_stages=-1, norm_eval=ne)
        if backbone:
            cfg.model["backbone"]["type"] = backbone
            if backbone in ("Resnet18", "Resnet34"):
                cfg.model["neck"]["in_channels"] = [64, 128, 256, 512]
        cfg.model["bbox_head"]["num_classes"] = classes + 1
        cfg.optimizer = dict(type='SGD', lr=lr or 0.0025, momentum=0.9, weight_decay=0.0001, grad_clip=dict(max_norm=35, norm_type=2))
        cfg.scheduler = dict(type='StepLR', warmup='linear', warmup_iters=50, warmup_ratio=1.0 / 3, milestones=[1000])
    else:
        cfg = Config(os.path.join(ROOT, "configs", "orcnn", "orcnn_van3_7_anchor.py"))
        cfg.model["backbone"] = dict(type="van_b0", img_size=tile, num_stages=4, out_indices=(0, 1, 2, 3))
        cfg.model["neck"]["in_channels"] = [32, 64, 160, 256]
        cfg.model["bbox_head"]["num_classes"] = classes
        cfg.optimizer = dict(type='AdamW', lr=lr or 0.0004, weight_decay=0.05)
        cfg.scheduler = dict(type='StepLR', warmup='linear', warmup_iters=50, warmup_ratio=1.0 / 3, milestones=[1000])
        cfg.optimizer_swa = cfg.scheduler_swa = None
        cfg.swa_start_epoch = None
    ds = dict(type="SyntheticDOTADataset", tile=tile, batch_size=batch, num_classes=classes, k_cycle=[4, 6, 5, 7],
              num_images=images, render=True, shuffle=True, min_size=(24, 12), max_size=(90, 40))
    cfg.dataset = dict(train=dict(ds), val=dict(ds, shuffle=False))
    cfg.max_epoch = 10 ** 6
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    mf = torch.channels_last if os.environ.get("MF", "") == "channels_last" else None      # (the bench's bf16 layout)
    r = Runner(cfg, device=dev, distributed=False, amp_dtype=torch.bfloat16 if dtype == "bf16" else None, memory_format=mf)
    r.build_datasets()
    return r


def train(r, iters, log=25):
    from rs_detection_amd.data import batch_to_device
    losses = []
    while len(losses) < iters:
        r.train_dataset.set_epoch(r.epoch)
        for images, targets in r.train_dataset:
            images, targets = batch_to_device(images, targets, r.device)
            total, parts = r.train_step(images, targets)
            losses.append(float(total.detach()))
            if log and len(losses) % log == 0:
                print("iter %4d loss %.4f (mean of last %d: %.4f)" % (len(losses), losses[-1], log, np.mean(losses[-log:])), flush=True)
            if len(losses) >= iters:
                break
        r.epoch += 1
    return losses


if __name__ == "__main__":
    if os.environ.get("ROUTES_OFF", "0") == "1":      # the per-operator routes of round 4 (A/B of the learning curve)
        import rs_detection_amd.ops.bottleneck as _b, rs_detection_amd.ops.conv_bn as _c, rs_detection_amd.ops.conv3x3 as _t
        _b._ON = _c._ON = _t._TOWER = False
    model = sys.argv[1] if len(sys.argv) > 1 else "s2anet"
    dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    r = make_runner(model, dtype, tile=int(os.environ.get("TILE", "256")), backbone=os.environ.get("BACKBONE"),
                    lr=float(os.environ["LR"]) if "LR" in os.environ else None)
    t0 = time.time()
    losses = train(r, iters)
    print("trained %d iters in %.1f s" % (iters, time.time() - t0))
    t0 = time.time()
    ev = r.val()
    print("mAP %.4f  (%.1f s)" % (ev["eval/0_meanAP"], time.time() - t0), {k: round(v, 3) for k, v in ev.items()})
