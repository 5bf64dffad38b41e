"""CPU: host-side logic on the path that needs no kernel: losses vs the oracle restatement,
LR schedule, ResNet freezing rules, ARF index table, anchor generator, synthetic stream."""
import math

import numpy as np
import pytest
import torch

import oracle


def test_focal_and_smooth_l1_vs_oracle():
    from rs_detection_amd.models.losses.focal_loss import FocalLoss
    from rs_detection_amd.models.losses.smooth_l1_loss import SmoothL1Loss
    rng = np.random.default_rng(0)
    pred = rng.standard_normal((400, 15)).astype(np.float32) * 3
    tgt = rng.integers(0, 16, 400).astype(np.int32)
    w = (rng.uniform(0, 1, 400) > 0.2).astype(np.float32)
    got = FocalLoss()(torch.from_numpy(pred), torch.from_numpy(tgt), torch.from_numpy(w), avg_factor=37.0)
    want = oracle.np_sigmoid_focal_loss(pred, tgt, w, 2.0, 0.25, 37.0)
    assert abs(float(got) - want) <= 1e-5 * max(1, abs(want))
    p, t = rng.standard_normal((300, 5)).astype(np.float32), rng.standard_normal((300, 5)).astype(np.float32)
    bw = (rng.uniform(0, 1, (300, 5)) > 0.5).astype(np.float32)
    got = SmoothL1Loss(beta=1 / 9.)(torch.from_numpy(p), torch.from_numpy(t), torch.from_numpy(bw), avg_factor=11.0)
    want = oracle.np_smooth_l1_loss(p, t, bw, 1 / 9., 11.0)
    assert abs(float(got) - want) <= 1e-5 * max(1, abs(want))


def test_parse_losses_and_multi_apply():
    from rs_detection_amd.utils.general import parse_losses, multi_apply, unmap
    total, parsed = parse_losses(dict(loss_a=[torch.tensor(1.0), torch.tensor(2.0)], loss_b=torch.tensor([3.0, 5.0]),
                                      acc=torch.tensor(9.0)))
    assert float(total) == 7.0 and set(parsed) == {"loss_a", "loss_b", "acc"}
    a, b = multi_apply(lambda x, y, k=0: (x + y + k, x * y), [1, 2], [3, 4], k=10)
    assert a == [14, 16] and b == [3, 8]
    m = torch.tensor([True, False, True])
    assert unmap(torch.tensor([5, 6]), 3, m).tolist() == [5, 0, 6]


def test_step_lr_with_linear_warmup():
    from rs_detection_amd.optims.optimizer import SGD
    from rs_detection_amd.optims.lr_scheduler import StepLR
    p = [torch.nn.Parameter(torch.zeros(3))]
    opt = SGD(p, lr=0.0025, momentum=0.9, weight_decay=1e-4, grad_clip=dict(max_norm=35, norm_type=2))
    sch = StepLR(optimizer=opt, milestones=[7, 10], warmup='linear', warmup_iters=500, warmup_ratio=1 / 3.)
    assert abs(opt.cur_lr() - 0.0025 / 3) < 1e-12          # lr_scheduler.py:30-36 at iter 0
    sch.step(250, 0)
    assert abs(opt.cur_lr() - 0.0025 * (1 - 0.5 * (2 / 3.))) < 1e-12
    sch.step(500, 0)
    assert abs(opt.cur_lr() - 0.0025) < 1e-12
    sch.step(9000, 7)
    assert abs(opt.cur_lr() - 0.00025) < 1e-12
    sch.step(9999, 10)
    assert abs(opt.cur_lr() - 0.000025) < 1e-12
    # grad clip: global L2 norm brought to 35
    p[0].grad = torch.tensor([300.0, 400.0, 0.0])
    opt._clip()
    assert abs(float(p[0].grad.norm()) - 35.0) < 1e-3


def test_resnet_freeze_and_norm_eval():
    from rs_detection_amd.models.backbones.resnet import Resnet50
    m = Resnet50(frozen_stages=1, return_stages=["layer1", "layer2", "layer3", "layer4"], pretrained=True)
    m.train()
    assert not any(p.requires_grad for p in m.conv1.parameters())
    assert not any(p.requires_grad for p in m.layer1.parameters())
    assert all(p.requires_grad for p in m.layer2.parameters())
    assert all(not b.training for b in m.modules() if isinstance(b, torch.nn.BatchNorm2d))  # q24
    frozen = sum(p.numel() for p in m.parameters() if not p.requires_grad)
    assert abs(frozen / 1e6 - 0.225) < 0.005
    assert m.pretrained_source == "jittorhub://resnet50.pkl"


def test_arf_index_table():
    from rs_detection_amd.ops.orn import arf_indices
    idx = arf_indices(1, 8, (3, 3))
    assert idx.shape == (1, 3, 3, 8) and idx.dtype == torch.uint8
    assert idx[0, :, :, 0].reshape(-1).tolist() == list(range(1, 10))          # 0 degrees = identity
    assert idx[0, :, :, 2].reshape(-1).tolist() == [3, 6, 9, 2, 5, 8, 1, 4, 7]  # 90 degrees (orn.py:658)
    for k in range(8):
        assert sorted(idx[0, :, :, k].reshape(-1).tolist()) == list(range(1, 10))  # permutations
    idx8 = arf_indices(8, 8, (1, 1))
    assert idx8[:, 0, 0, 1].tolist() == [2, 3, 4, 5, 6, 7, 8, 1]                # orientation shift


def test_anchor_generator_matches_closed_form():
    from rs_detection_amd.models.boxes.anchor_generator import AnchorGeneratorRotatedS2ANet
    for s in (8, 32, 128):
        g = AnchorGeneratorRotatedS2ANet(s, [4], [1.0])
        f = 1024 // s
        a = g.grid_anchors((f, f), s).numpy()
        assert (a == oracle.np_s2anet_grid_anchors((f, f), s)).all()
        v = g.valid_flags((4, 6), (3, 5))
        assert v.view(4, 6).sum().item() == 15 and not v.view(4, 6)[3].any() and not v.view(4, 6)[:, 5].any()


def test_align_conv_get_offset_reference_signature():
    """Pure-torch AlignConv.get_offset (reference signature) vs the NumPy transcription."""
    from rs_detection_amd.models.roi_heads.s2anet_head import AlignConv
    rng = np.random.default_rng(1)
    anchors = oracle.np_s2anet_grid_anchors((6, 5), 16)
    anchors[:, 2:4] *= rng.uniform(0.5, 2, (30, 2)).astype(np.float32)
    anchors[:, 4] = rng.uniform(-0.7, 2.3, 30).astype(np.float32)
    got = AlignConv(4, 4).get_offset(torch.from_numpy(anchors), (6, 5), 16).numpy()
    np.testing.assert_allclose(got, oracle.np_align_conv_offset(anchors, (6, 5), 16), atol=1e-5)


def test_synthetic_stream_is_deterministic_and_dota_shaped():
    from rs_detection_amd.utils import synthetic as syn
    a, b = syn.synthetic_targets(4, rank=1, it=3), syn.synthetic_targets(4, rank=1, it=3)
    assert [t["rboxes"].shape[0] for t in syn.synthetic_targets(4)] == [16, 100, 400, 40]
    for x, y in zip(a, b):
        assert (x["rboxes"] == y["rboxes"]).all() and (x["labels"] == y["labels"]).all()
        r = x["rboxes"]
        assert (r[:, 2] >= r[:, 3]).all() and (r[:, 4] >= -math.pi / 4).all() and (r[:, 4] < 3 * math.pi / 4).all()
        assert x["labels"].min() >= 1 and x["labels"].max() <= 15
    assert not (a[0]["rboxes"] == syn.synthetic_targets(4, rank=0, it=3)[0]["rboxes"]).all()  # ranks differ
    assert syn.s2anet_anchor_grid().shape == (21824, 5)


def test_xcd_workgroup_renumbering_is_a_bijection():
    """The two XCD-aware workgroup renumberings of csrc/rsdet_api_internal.h (rsdet_xcd_contiguous, rsdet_xcd_band),
    restated in Python: every logical item is produced exactly once, ids that share an XCD (equal id % 8) map to one
    contiguous run / band -- for totals that are and are not multiples of 8."""
    def contiguous(i, total):
        q, r, xcd, slot = total >> 3, total & 7, i & 7, i >> 3
        return xcd * q + min(xcd, r) + slot

    for total in (1, 7, 8, 9, 63, 64, 1000, 4097):
        out = [contiguous(i, total) for i in range(total)]
        assert sorted(out) == list(range(total))
        for xcd in range(8):
            mine = sorted(out[i] for i in range(total) if i % 8 == xcd)
            assert mine == list(range(mine[0], mine[0] + len(mine))) if mine else True

    def band(i, n_outer, n_inner):
        xcd, slot = i & 7, i >> 3
        o0, o1 = n_outer * xcd // 8, n_outer * (xcd + 1) // 8
        ln = o1 - o0
        if ln <= 0 or slot >= ln * n_inner:
            return None
        inner = slot // ln
        return (o0 + slot - inner * ln, inner)

    for n_outer, n_inner in ((1, 1), (5, 3), (8, 16), (13, 7), (128, 16), (257, 2)):
        grid = 8 * ((n_outer + 7) // 8) * n_inner
        items = [band(i, n_outer, n_inner) for i in range(grid)]
        got = sorted(x for x in items if x is not None)
        assert got == sorted((o, k) for o in range(n_outer) for k in range(n_inner))
        for xcd in range(8):
            outers = sorted({x[0] for i, x in enumerate(items) if x is not None and i % 8 == xcd})
            assert outers == list(range(outers[0], outers[0] + len(outers))) if outers else True


def test_packaged_miopen_records_match_rule(monkeypatch, tmp_path):
    """bench.py runs the fp32 step in channels_last only when the packaged find records belong to the MIOpen in use and
    are the ones MIOpen will read (utils/miopen_db.packaged_records_match): version taken from the record file names."""
    import torch
    from rs_detection_amd.utils import miopen_db as m
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.delenv("RSDET_NO_MIOPEN_DB", raising=False)
    monkeypatch.setattr(torch.backends.cudnn, "version", lambda: 3005000)
    assert m.packaged_records_match()
    monkeypatch.setattr(torch.backends.cudnn, "version", lambda: 3006001)      # another MIOpen: its own file names
    assert not m.packaged_records_match()
    monkeypatch.setattr(torch.backends.cudnn, "version", lambda: 3005000)
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", str(tmp_path))                   # the user's own database wins
    assert not m.packaged_records_match()
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    monkeypatch.setenv("RSDET_NO_MIOPEN_DB", "1")
    assert not m.packaged_records_match()
