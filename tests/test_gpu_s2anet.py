"""GPU: S2ANet end to end (a5/a6/a14/a15).
 * anchor targets: the batched device form the head uses == the reference-shaped per-image form == a NumPy twin
   built from the oracle (IoU from the reference CPU source restatement, assigner.py:111-170, PseudoSampler,
   bbox2delta_rotated box_ops.py:184-236, unmap) -- labels / weights exact, regression targets to 1e-5;
 * the model from the config file: fp32 train steps, bf16-autocast train step, eval output format."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import dota_boxes, s2anet_anchors

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAM = dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0, ignore_iof_thr=-1,
                         iou_calculator=dict(type='BboxOverlaps2D_rotated')),
           bbox_coder=dict(type='DeltaXYWHABBoxCoder', target_means=(0., 0., 0., 0., 0.), target_stds=(1., 1., 1., 1., 1.),
                           clip_border=True),
           allowed_border=-1, pos_weight=-1, debug=False)


def _np_anchor_target(oc, anchors, gts, labels):
    """anchor_target.py:97-170 for one image, sampling=False, reg_decoded_bbox=False."""
    A = anchors.shape[0]
    lab = np.zeros(A, np.int32)
    lw = np.zeros(A, np.float32)
    bt = np.zeros((A, 5), np.float32)
    bw = np.zeros((A, 5), np.float32)
    if gts.shape[0] == 0:
        lw[:] = 1.0                       # assigner.py:125-131: no gt -> everything negative
        return lab, lw, bt, bw, 0, A
    ov = oc.box_iou_rotated(gts, anchors, 0)
    gi, _, al = oc.assign_wrt_overlaps(ov, 0.5, 0.4, 0.0, True, True, labels, 0)
    pos, neg = np.nonzero(gi > 0)[0], np.nonzero(gi == 0)[0]
    if len(pos):
        bt[pos] = oracle.np_bbox2delta_rotated(anchors[pos], gts[gi[pos] - 1])
        bw[pos] = 1.0
        lab[pos] = labels[gi[pos] - 1]
        lw[pos] = 1.0
    lw[neg] = 1.0
    return lab, lw, bt, bw, len(pos), len(neg)


@pytest.mark.parametrize("ks", [[16, 100, 0, 40], [1], [400, 3]])
def test_anchor_target_batched_vs_reference_shaped_vs_numpy(cuda, oracle_c, ks):
    import rs_detection_amd.models  # noqa: F401
    from rs_detection_amd.models.boxes.anchor_target import anchor_target, anchor_target_batched
    rng = np.random.default_rng(sum(ks) + len(ks))
    anchors = s2anet_anchors()
    A = anchors.shape[0]
    gts = [dota_boxes(rng, k) if k else np.zeros((0, 5), np.float32) for k in ks]
    labs = [rng.integers(1, 16, k).astype(np.int32) for k in ks]
    # refined-anchor case as well: per-image anchors = grid perturbed
    ref = np.stack([anchors + np.concatenate([rng.normal(0, 4, (A, 2)), np.zeros((A, 3))], 1).astype(np.float32) for _ in ks])
    ref[:, :, 2:4] *= np.exp(rng.normal(0, 0.2, (len(ks), A, 2))).astype(np.float32)
    ref[:, :, 4] += rng.normal(0, 0.3, (len(ks), A)).astype(np.float32)
    for per_image, fused in ((False, False), (True, False), (False, True), (True, True)):
        anc_np = ref if per_image else np.broadcast_to(anchors, (len(ks), A, 5))
        t = lambda a: torch.from_numpy(np.array(a)).to(cuda)
        ro = torch.tensor(np.concatenate([[0], np.cumsum(ks)]), dtype=torch.int32, device=cuda)
        # fused = the sparse two-launch path the head runs (ks known on the host); otherwise the dense chain
        got = anchor_target_batched(t(ref) if per_image else t(anchors), t(np.concatenate(gts)), t(np.concatenate(labs)),
                                    ro, max(max(ks), 1), FAM, ks=ks if fused else None,
                                    heavy_from=20480 if fused else None)
        labels, lw, bt, bw, npos, nneg = [g.cpu().numpy() for g in got]
        # reference-shaped per-image form (single pyramid "level" holding all anchors).  Like the reference
        # (assigner.py:92-93) it raises on an image without gts -- the batched form treats it as all-negative.
        metas = [dict(img_shape=(1024, 1024, 3), pad_shape=(1024, 1024, 3)) for _ in ks]
        full = [i for i, k in enumerate(ks) if k > 0]
        per = anchor_target([[t(anc_np[i])] for i in full],
                            [[torch.ones(A, dtype=torch.bool, device=cuda)] for _ in full],
                            [t(gts[i]) for i in full], [metas[i] for i in full], (0.,) * 5, (1.,) * 5, FAM,
                            gt_labels_list=[t(labs[i]) for i in full], sampling=False)
        if len(full) < len(ks):
            with pytest.raises(ValueError):
                anchor_target([[t(anc_np[i])] for i in range(len(ks))],
                              [[torch.ones(A, dtype=torch.bool, device=cuda)] for _ in ks], [t(g) for g in gts], metas,
                              (0.,) * 5, (1.,) * 5, FAM, gt_labels_list=[t(l) for l in labs], sampling=False)
        want_pos = want_neg = per_pos = per_neg = 0
        for i in range(len(ks)):
            wl, wlw, wbt, wbw, p, q = _np_anchor_target(oracle_c, np.ascontiguousarray(anc_np[i]), gts[i], labs[i])
            want_pos += max(p, 1)
            want_neg += max(q, 1)
            assert (labels[i] == wl).all(), (i, int((labels[i] != wl).sum()))
            assert (lw[i] == wlw).all() and (bw[i] == wbw).all()
            np.testing.assert_allclose(bt[i], wbt, rtol=1e-5, atol=1e-5)
            if i in full:
                j = full.index(i)
                per_pos += max(p, 1)
                per_neg += max(q, 1)
                assert (per[0][0][j].cpu().numpy() == wl).all() and (per[1][0][j].cpu().numpy() == wlw).all()
                np.testing.assert_allclose(per[2][0][j].cpu().numpy(), wbt, rtol=1e-5, atol=1e-5)
                assert (per[3][0][j].cpu().numpy() == wbw).all()
        assert int(npos) == want_pos and int(nneg) == want_neg      # sum_img max(#,1)  (q11)
        assert per[4] == per_pos and per[5] == per_neg


def _runner(cuda, amp=None):
    from rs_detection_amd.config import Config
    from rs_detection_amd.runner.runner import Runner
    cfg = Config(os.path.join(ROOT, "configs", "s2anet", "s2anet_r50_fpn_1x_dota.py"))
    torch.manual_seed(0)
    return Runner(cfg, device=cuda, distributed=False, amp_dtype=amp)


def _batch(cuda, n=2, size=256, k=20):
    from rs_detection_amd.utils import synthetic as syn
    images = torch.randn(n, 3, size, size, device=cuda)
    targets = []
    for t in syn.synthetic_targets(n, img=size):
        t = dict(t)
        t["rboxes"] = torch.from_numpy(t["rboxes"][:k]).to(cuda)
        t["labels"] = torch.from_numpy(t["labels"][:k]).to(cuda)
        targets.append(t)
    return images, targets


def test_s2anet_train_steps_fp32_and_eval_format(cuda):
    runner = _runner(cuda)
    images, targets = _batch(cuda)
    losses = []
    for _ in range(3):
        total, parsed = runner.train_step(images, targets)
        losses.append(float(total))
        assert set(parsed) >= {"loss_fam_cls", "loss_fam_bbox", "loss_odm_cls", "loss_odm_bbox"}
    assert all(np.isfinite(losses))
    res = runner.predict(images, targets)
    assert len(res) == 2
    for polys, scores, labels in res:                      # s2anet_head.py:624-629
        assert polys.dim() == 2 and polys.shape[1] == 8 and polys.shape[0] == scores.shape[0] == labels.shape[0]


def test_s2anet_bf16_step_tracks_fp32_with_normalised_activations(cuda):
    """The tight form of the bf16-vs-fp32 check: with BatchNorm in training mode (what a from-scratch run uses) the
    activations stay normalised, bf16 round-off is not amplified through the 50 layers, and all four first-step losses
    of the bf16 autocast step (NCHW and channels_last, implicit-GEMM AlignConv, fused losses / targets) agree with the
    fp32 step within a few percent -- against 15-30 % in the eval-mode test below."""
    from rs_detection_amd.config import Config
    from rs_detection_amd.runner.runner import Runner
    images, targets = _batch(cuda)

    def first_step(amp, mf=None):
        cfg = Config(os.path.join(ROOT, "configs", "s2anet", "s2anet_r50_fpn_1x_dota.py"))
        cfg.model["backbone"].update(pretrained=False, frozen_stages=-1, norm_eval=False)
        torch.manual_seed(0)
        r = Runner(cfg, device=cuda, distributed=False, amp_dtype=amp, memory_format=mf)
        _, parsed = r.train_step(images, targets)
        return {k: float(v) for k, v in parsed.items()}

    ref = first_step(None)
    for mf in (None, torch.channels_last):
        got = first_step(torch.bfloat16, mf)
        # measured over repeated runs (scratch/bf16_track.py): <= 0.1 % on the classification losses, 0.3-0.4 % on the
        # FAM regression loss, 0.6-2.9 % on the ODM regression loss -- that one moves in steps, run to run, because the
        # ODM targets are ASSIGNED on the refined anchors and a borderline IoU flips with bf16 round-off: 10 % for it
        for k, tol in (("loss_fam_cls", 0.05), ("loss_odm_cls", 0.05), ("loss_fam_bbox", 0.05), ("loss_odm_bbox", 0.10)):
            assert abs(got[k] - ref[k]) / abs(ref[k]) < tol, (mf, k, got[k], ref[k])


def test_s2anet_train_step_bf16_autocast(cuda):
    """configs[2]/[4]: bf16 autocast over the MIOpen / rocBLAS part and AlignConv (bf16 matrix cores, fp32 accumulate);
    ARF / RROIAlign / the box kernels stay fp32 (custom_fwd(cast_inputs=float32))."""
    runner = _runner(cuda, torch.bfloat16)
    images, targets = _batch(cuda)
    for _ in range(2):
        total, _ = runner.train_step(images, targets)
        assert np.isfinite(float(total))
    # same init, same batch.  Loose on purpose: at random init with BatchNorm in eval mode (identity statistics) the
    # 50-layer backbone amplifies bf16 round-off to ~14 % relative error on the FPN outputs (measured), which moves
    # the regression losses by ~15 %; the classification losses (prior-bias dominated) stay within a few percent.
    _, ref = _runner(cuda).train_step(images, targets)
    _, got = _runner(cuda, torch.bfloat16).train_step(images, targets)
    for k, tol in (("loss_fam_cls", 0.05), ("loss_odm_cls", 0.05), ("loss_fam_bbox", 0.3), ("loss_odm_bbox", 0.3)):
        assert abs(float(got[k]) - float(ref[k])) / abs(float(ref[k])) < tol, (k, float(got[k]), float(ref[k]))


def test_dota_dataset_feeds_the_model_and_the_map_driver(cuda, tmp_path):
    """SURVEY 8f rank 2 + 1 end to end: labels.pkl + images -> DOTADataset (PIL transforms) -> batch -> S2ANet train
    step -> eval -> DOTA mAP with the polygon overlaps on the GPU."""
    from test_data_pipeline_cpu import _make_dataset
    import rs_detection_amd.data as D
    from rs_detection_amd.utils.registry import DATASETS, build_from_cfg
    _make_dataset(tmp_path, n=5)
    ds = build_from_cfg(dict(type="DOTADataset", dataset_dir=str(tmp_path), batch_size=2, shuffle=False,
                             transforms=[dict(type="RotatedResize", min_size=256, max_size=256),
                                         dict(type="RotatedRandomFlip", prob=0.5), dict(type="Pad", size_divisor=32),
                                         dict(type="Normalize", mean=[123.675, 116.28, 103.53],
                                              std=[58.395, 57.12, 57.375], to_bgr=False)]), DATASETS)
    runner = _runner(cuda)
    results = []
    for images, targets in ds:
        timg, ttg = D.batch_to_device(images, targets, cuda)
        total, _ = runner.train_step(timg, ttg)
        assert np.isfinite(float(total))
        preds = runner.predict(timg, ttg)
        for (polys, scores, labels), t in zip(preds, targets):
            results.append(((polys.cpu().numpy().astype(np.float64), scores.cpu().numpy(), labels.cpu().numpy().astype(np.int64)), t))
    aps = ds.evaluate(results, device=cuda)
    assert "eval/0_meanAP" in aps and 0.0 <= aps["eval/0_meanAP"] <= 1.0


def test_runner_epoch_loop_checkpoint_and_eval(cuda, tmp_path):
    """The reference's run loop (runner.py:91-103) on a generated DOTA-format folder: two epochs, a checkpoint per
    epoch in the reference's pickle layout, mAP after each epoch, then resume into a fresh runner."""
    from test_data_pipeline_cpu import _make_dataset
    from rs_detection_amd.config import Config
    from rs_detection_amd.runner.runner import Runner
    data = tmp_path / "data"
    data.mkdir()
    _make_dataset(data, n=5)
    tf = [dict(type="RotatedResize", min_size=256, max_size=256), dict(type="Pad", size_divisor=32),
          dict(type="Normalize", mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_bgr=False)]
    cfg = Config(os.path.join(ROOT, "configs", "s2anet", "s2anet_r50_fpn_1x_dota.py"))
    cfg.dataset = dict(train=dict(type="DOTADataset", dataset_dir=str(data), batch_size=2, shuffle=True, transforms=tf),
                       val=dict(type="DOTADataset", dataset_dir=str(data), batch_size=2, transforms=tf))
    cfg.max_epoch, cfg.checkpoint_interval, cfg.eval_interval = 2, 1, 1
    torch.manual_seed(0)
    r = Runner(cfg, device=cuda, distributed=False).build_datasets(work_dir=str(tmp_path / "work"))
    evals = r.run(log_interval=0)
    assert r.epoch == 2 and r.iter == 4 and set(evals) == {1, 2}
    assert all("eval/0_meanAP" in e for e in evals.values())
    ck = tmp_path / "work" / "checkpoints"
    assert sorted(os.listdir(ck)) == ["ckpt_1.pkl", "ckpt_2.pkl"]
    torch.manual_seed(1)
    r2 = Runner(cfg, device=cuda, distributed=False)
    loaded, missing, unexpected, mismatched = r2.load(str(ck / "ckpt_2.pkl"))
    assert not missing and not unexpected and not mismatched and (r2.epoch, r2.iter) == (2, 4)
    for k, v in r.model.state_dict().items():
        assert torch.equal(v, r2.model.state_dict()[k]), k


def test_runner_test_flip_and_submission(cuda, tmp_path):
    """runner.py:210-249 on a folder of tiles: predictions of the plain and the H / V flipped passes, the pickle, the
    per-class Task-1 files before and after the tile merge (polygon NMS on the GPU) and the submission zip."""
    import pickle
    import zipfile
    from PIL import Image
    from rs_detection_amd.config import Config
    from rs_detection_amd.runner.runner import Runner
    tiles = tmp_path / "tiles"
    tiles.mkdir()
    rng = np.random.default_rng(0)
    names = ["P0001__1.0__0___0", "P0001__1.0__200___0", "P0002__0.5__0___0"]
    for n in names:
        Image.fromarray(rng.integers(0, 255, (256, 256, 3), dtype=np.uint8)).save(tiles / (n + ".png"))
    cfg = Config(os.path.join(ROOT, "configs", "s2anet", "s2anet_r50_fpn_1x_dota.py"))
    tf = [dict(type="RotatedResize", min_size=256, max_size=256), dict(type="Pad", size_divisor=32),
          dict(type="Normalize", mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_bgr=False)]
    cfg.dataset = dict(test=dict(type="ImageDataset", images_dir=str(tiles), transforms=tf, dataset_type="DOTA",
                                 batch_size=2))
    cfg.flip_test = ["H", "V"]
    cfg.model["bbox_head"]["test_cfg"]["score_thr"] = 0.0       # random weights: keep something to merge
    cfg.model["bbox_head"]["test_cfg"]["max_per_img"] = 40
    torch.manual_seed(0)
    r = Runner(cfg, device=cuda, distributed=False).build_datasets(work_dir=str(tmp_path / "work"))
    out = r.test(name="sub")
    res = pickle.load(open(out["pkl"], "rb"))
    assert len(res) == 3 * 3                                     # 3 tiles x (plain, H, V)
    assert sorted(t.get("flip_mode", "") for _, t in res) == [""] * 3 + ["H"] * 3 + ["V"] * 3
    (polys, scores, labels), t0 = res[0]
    assert polys.shape[1] == 8 and len(scores) == len(labels) == polys.shape[0] > 0
    before, after = (os.path.join(str(tmp_path / "work"), "test", "submit_0", d) for d in ("before_nms", "after_nms"))
    assert os.listdir(before) and sorted(os.listdir(after)) == sorted(os.listdir(before))
    merged = open(os.path.join(after, sorted(os.listdir(after))[0])).read().split()
    assert merged[0] in ("P0001", "P0002")                      # back in whole-image names / coordinates
    n_before = sum(len(open(os.path.join(before, f)).readlines()) for f in os.listdir(before))
    n_after = sum(len(open(os.path.join(after, f)).readlines()) for f in os.listdir(after))
    assert 0 < n_after < n_before                               # the three passes of a tile overlap: NMS removed some
    with zipfile.ZipFile(out["submission"]) as z:
        assert sorted(z.namelist()) == sorted(os.listdir(after))


def test_s2anet_bf16_channels_last_step_tracks_the_nchw_step(cuda):
    """The bf16 line runs channels_last (NHWC BatchNorm tails, ops/bn_act.py; MIOpen's bf16 kernels are NHWC-native):
    same model, same batch -> the same losses as the NCHW bf16 step to bf16 round-off, and finite gradients."""
    from rs_detection_amd.config import Config
    from rs_detection_amd.runner.runner import Runner
    cfg = Config(os.path.join(ROOT, "configs", "s2anet", "s2anet_r50_fpn_1x_dota.py"))
    images, targets = _batch(cuda)
    out = {}
    for name, mf in (("nchw", None), ("cl", torch.channels_last), ("trunk", "trunk_channels_last")):
        torch.manual_seed(0)
        r = Runner(cfg, device=cuda, distributed=False, amp_dtype=torch.bfloat16, memory_format=mf)
        _, parsed = r.train_step(images, targets)
        out[name] = {k: float(v) for k, v in parsed.items()}
        assert all(np.isfinite(v) for v in out[name].values())
        assert all(torch.isfinite(p.grad).all() for p in r.model.parameters() if p.grad is not None)
    for name in ("cl", "trunk"):
        for k, tol in (("loss_fam_cls", 0.05), ("loss_odm_cls", 0.05), ("loss_fam_bbox", 0.3), ("loss_odm_bbox", 0.3)):
            assert abs(out[name][k] - out["nchw"][k]) / abs(out["nchw"][k]) < tol, (name, k, out[name][k], out["nchw"][k])
