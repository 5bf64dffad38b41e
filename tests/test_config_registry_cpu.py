"""CPU: JDet config loader + registry semantics (mirrors the reference's tests/test_config cases
on this repo's own fixture files) and the S2ANet config against the reference's, when mounted."""
import os

import pytest

from rs_detection_amd.config import Config, init_cfg, get_cfg
from rs_detection_amd.utils.registry import Registry, build_from_cfg

CFG = os.path.join(os.path.dirname(__file__), "golden", "config")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dump(name):
    init_cfg(os.path.join(CFG, name))
    d = get_cfg().dump()
    assert d.pop("name") == os.path.splitext(os.path.basename(name))[0]  # config.py:107-110
    assert d.pop("work_dir").startswith("work_dirs/")
    return d


CASES = {
    "plain.yaml": {'alpha': 7, 'net': {'depth': 50, 'width': 64}},
    "child.yaml": {'alpha': 7, 'net': {'depth': 101, 'width': 64, 'stages': 4}},            # _base_
    "cover.yaml": {'alpha': 7, 'net': {'depth': 18}},                                          # _cover_
    "root_cover.yaml": {'net': {'depth': 34, 'extra': {}}, 'opt': {'steps': [8, 11]}},        # root _cover_
    "sub/leaf.yaml": {'alpha': 7, 'net': {'depth': 152, 'width': 64}, 'beta': 2},             # sub-directory bases
    "tree.yaml": {'alpha': 7, 'net': {'depth': 50, 'width': 96, 'head': {'convs': 4}}, 'gamma': 0},  # tree of bases
    "simple.py": {'lr': 0.0025, 'epochs': 12},                                                 # .py config
    "with_base.py": {'alpha': 7, 'net': {'depth': 50, 'width': 64}, 'epochs': 36},            # .py with yaml base
    "computed.py": {'strides': [8, 16, 32, 64, 128], 'levels': 5, 'tag': 's8', 'out': 's8/ckpt'},  # computation, module dropped
    "mix/top.py": {'fresh': 2, 'kept': 3, 'sched': {'warm': 500, 'kind': 'step'},
                   'net': {'depth': 50, 'gone': -2, 'width': 64}, 'alpha': 7, 'omega': 9},     # mixture
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_config_cases(name):
    assert dump(name) == CASES[name]


def test_config_attribute_access_and_missing_key():
    cfg = Config(os.path.join(CFG, "plain.yaml"))
    assert cfg.alpha == 7 and cfg.net.depth == 50 and cfg.net == {'depth': 50, 'width': 64}
    assert cfg.not_there is None and cfg.net.nope is None  # config.py:24-27 (q23)


def test_registry_semantics():
    R = Registry()

    @R.register_module()
    class A:
        def __init__(self, x=1, y=2):
            self.x, self.y = x, y

    R.register_module(name="B", module=dict)
    assert build_from_cfg(dict(type='A', x=5), R, y=7).__dict__ == {'x': 5, 'y': 7}
    assert isinstance(build_from_cfg('A', R), A)
    assert build_from_cfg(None, R) is None
    with pytest.raises(AssertionError):
        R.register_module(name="A", module=A)  # duplicate
    with pytest.raises(AssertionError):
        R.get("missing")
    with pytest.raises(TypeError, match="A"):
        build_from_cfg(dict(type='A', z=1), R)  # re-raised with the class name (registry.py:34-39)
    with pytest.raises(TypeError):
        build_from_cfg(3, R)


def test_all_registries_present_and_populated():
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.optims import optimizer, lr_scheduler  # noqa: F401
    from rs_detection_amd.utils import registry as r
    for n in ("DATASETS TRANSFORMS MODELS BACKBONES HEADS LOSSES OPTIMS BRICKS NECKS SCHEDULERS BOXES HOOKS "
              "ROI_EXTRACTORS SHARED_HEADS").split():
        assert isinstance(getattr(r, n), r.Registry)
    for reg, names in ((r.MODELS, ["S2ANet"]), (r.BACKBONES, ["Resnet50", "Resnet101"]), (r.NECKS, ["FPN"]),
                       (r.HEADS, ["S2ANetHead"]), (r.LOSSES, ["FocalLoss", "SmoothL1Loss"]),
                       (r.BOXES, ["MaxIoUAssigner", "BboxOverlaps2D_rotated", "BboxOverlaps2D_rotated_v1",
                                  "BboxOverlaps2D", "DeltaXYWHABBoxCoder", "AnchorGeneratorRotatedS2ANet"]),
                       (r.OPTIMS, ["SGD", "AdamW"]), (r.SCHEDULERS, ["StepLR", "CosineAnnealingLR"])):
        for n in names:
            assert n in reg, n


def test_s2anet_config_builds_and_matches_reference():
    import torch
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.utils.registry import MODELS
    mine = Config(os.path.join(ROOT, "configs", "s2anet", "s2anet_r50_fpn_1x_dota.py"))
    model = build_from_cfg(mine.model, MODELS)
    trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert abs(trainable / 1e6 - 36.2) < 0.05  # 36.2 M trainable parameters
    sd = model.state_dict()
    for k in ("backbone.layer1.0.conv1.weight", "neck.lateral_convs.0.conv.weight", "neck.fpn_convs.4.conv.weight",
              "bbox_head.fam_reg_convs.0.conv.weight", "bbox_head.align_conv.deform_conv.weight",
              "bbox_head.or_conv.weight", "bbox_head.or_pool.conv.0.weight", "bbox_head.odm_cls.bias"):
        assert k in sd, k
    assert tuple(sd["bbox_head.or_conv.weight"].shape) == (32, 256, 1, 3, 3)
    assert tuple(sd["bbox_head.or_conv.bias"].shape) == (256,)                # q15
    assert abs(float(sd["bbox_head.odm_cls.bias"][0]) + 4.59512) < 1e-4       # -log(99)
    ref = "/root/reference/configs/s2anet/s2anet_r50_fpn_1x_dota.py"
    if os.path.exists(ref):  # build container only: the reference's own file loads unchanged and agrees
        theirs = Config(ref)
        for k in ("model", "optimizer", "scheduler", "max_epoch", "log_interval", "checkpoint_interval"):
            assert mine.dump()[k] == theirs.dump()[k], k
    # backbone + FPN run on CPU tensors (torch); the head then needs the HIP ops and must refuse
    model.train()
    x = torch.randn(1, 3, 64, 64)
    feats = model.neck(model.backbone(x))
    assert [tuple(f.shape[-2:]) for f in feats] == [(8, 8), (4, 4), (2, 2), (1, 1), (1, 1)]
    from rs_detection_amd import _lib
    with pytest.raises(_lib.RsdetError):
        model(x, [dict(rboxes=torch.tensor([[32., 32, 20, 10, 0]]), labels=torch.tensor([1]), rboxes_ignore=None,
                       img_size=(64, 64), pad_shape=(64, 64), scale_factor=1.0)])
