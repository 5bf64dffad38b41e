"""GPU: the one-launch rotated IoU (rsdet_box_iou_rotated_tiled_f32) and the fused sparse anchor targets
(rsdet_anchor_target_rotated_f32) against
  * the oracle (restatement of the reference CPU source, pinned bit for bit by oracle/_ref) -- IoU within 1e-4,
    exact zeros, assignment indices exact;
  * the round-1 dense chain (grouped IoU -> assign_wrt_overlaps -> bbox2delta), which they must equal BIT FOR BIT:
    same clipper, same thresholds, same coder arithmetic."""
import numpy as np
import pytest
import torch

import oracle
from conftest import dota_boxes, s2anet_anchors, degenerate_boxes

pytestmark = pytest.mark.gpu


def _ro(ks, dev):
    return torch.tensor(np.concatenate([[0], np.cumsum(ks)]), dtype=torch.int32, device=dev)


def _refined(rng, anchors, n):
    A = anchors.shape[0]
    ref = np.stack([anchors.copy() for _ in range(n)])
    ref[:, :, :2] += rng.normal(0, 4, (n, A, 2)).astype(np.float32)
    ref[:, :, 2:4] *= np.exp(rng.normal(0, 0.2, (n, A, 2))).astype(np.float32)
    ref[:, :, 4] += rng.normal(0, 0.3, (n, A)).astype(np.float32)
    return ref.astype(np.float32)


@pytest.mark.parametrize("version", [0, 1])
@pytest.mark.parametrize("ks,per_image,with_table", [([16, 100, 400, 40], False, True), ([7, 33], True, True),
                                                     ([1], False, False), ([40, 0, 17], True, False)])
def test_tiled_iou_equals_three_launch_form_and_oracle(cuda, oracle_c, version, ks, per_image, with_table):
    from rs_detection_amd import ops
    rng = np.random.default_rng(sum(ks) + version)
    anchors = s2anet_anchors()
    A = anchors.shape[0]
    gts = np.concatenate([dota_boxes(rng, k) for k in ks]) if sum(ks) else np.zeros((0, 5), np.float32)
    b2 = _refined(rng, anchors, len(ks)) if per_image else anchors
    t1, t2, ro = torch.from_numpy(gts).to(cuda), torch.from_numpy(b2).to(cuda), _ro(ks, cuda)
    want = ops.box_iou_rotated_grouped(t1, ro, max(max(ks), 1), t2, version)
    got = ops.box_iou_rotated_tiled(t1, t2, ro, ks=ks if with_table else None, max_rows=max(ks), version=version)
    assert torch.equal(got, want)                      # one launch: bit for bit, zeros included
    two = ops.box_iou_rotated_tiled(t1, t2, ro, ks=ks if with_table else None, max_rows=max(ks), version=version,
                                    split=True)        # detection | zero fill + balanced clip
    assert torch.equal(two, want)
    # the same with the tiles of the large anchors (pyramid levels 2..4) cut into 4-row sub-tiles
    for split in (True, False):
        hv = ops.box_iou_rotated_tiled(t1, t2, ro, ks=ks if with_table else None, max_rows=max(ks), version=version,
                                       prepared=ops.prepare_boxes(t2, heavy_from=20480), split=split)
        assert torch.equal(hv, want)
    # oracle on one image (restatement of the reference CPU loop)
    g = int(np.argmax(ks))
    r0 = int(np.sum(ks[:g]))
    ref = oracle_c.box_iou_rotated(gts[r0:r0 + ks[g]], np.ascontiguousarray(b2[g] if per_image else b2), version)
    mine = got[r0:r0 + ks[g]].cpu().numpy()
    assert np.abs(mine - ref).max() <= 1e-4
    assert ((mine == 0) == (ref == 0)).mean() > 0.99999


def test_tiled_iou_plain_form_degenerate_and_ragged(cuda, oracle_c):
    """n2 not a multiple of 256 / 64 / 2, rows not a multiple of 16, touching / nested / zero-area boxes."""
    from rs_detection_amd import ops
    rng = np.random.default_rng(5)
    d = degenerate_boxes()
    for n1, n2 in ((15, 15), (17, 321), (3, 1), (40, 1027)):
        b1 = np.concatenate([d, dota_boxes(rng, max(n1 - len(d), 0), 200)])[:n1]
        b2 = np.concatenate([d, dota_boxes(rng, max(n2 - len(d), 0), 200)])[:n2]
        want = ops.box_iou_rotated(torch.from_numpy(b1).to(cuda), torch.from_numpy(b2).to(cuda))
        for split in (True, False):
            got = ops.box_iou_rotated_tiled(torch.from_numpy(b1).to(cuda), torch.from_numpy(b2).to(cuda), split=split)
            assert torch.equal(got, want), (n1, n2, split)
        assert np.abs(got.cpu().numpy() - oracle_c.box_iou_rotated(b1, b2, 0)).max() <= 1e-4
    # per-group slabs with an odd number of columns (slab pitch is padded to keep 16-byte alignment)
    ks = [5, 9]
    b1 = dota_boxes(rng, 14, 100)
    b2 = np.stack([dota_boxes(rng, 77, 100), dota_boxes(rng, 77, 100)])
    ro = _ro(ks, cuda)
    got = ops.box_iou_rotated_tiled(torch.from_numpy(b1).to(cuda), torch.from_numpy(b2).to(cuda), ro, ks=ks)
    want = ops.box_iou_rotated_grouped(torch.from_numpy(b1).to(cuda), ro, 9, torch.from_numpy(b2).to(cuda))
    assert torch.equal(got, want)


def _dense_chain(ops, anchors_t, gt_t, lab_t, ro, ks, valid=None, version=0, means=None, stds=None):
    """Round-1 path: grouped IoU -> (valid mask) -> assign kernel -> torch glue of anchor_target_batched."""
    B, A = len(ks), anchors_t.shape[-2]
    ov = ops.box_iou_rotated_grouped(gt_t, ro, max(max(ks), 1), anchors_t, version)
    if valid is not None:
        rows = torch.arange(ov.shape[0], device=ov.device)
        grp = torch.bucketize(rows, ro[1:].long(), right=True).clamp(max=B - 1)
        ov = torch.where(valid[grp], ov, ov.new_tensor(-1.0))
    gi, mo, lab = ops.assign_wrt_overlaps(ov, ro, max(max(ks), 1), 0.5, 0.4, 0.0, True, True, lab_t, 0)
    pos, neg = gi > 0, gi == 0
    gidx = ((gi.long() - 1).clamp(min=0) + ro[:-1].long()[:, None]).clamp(max=max(gt_t.shape[0] - 1, 0))
    anc = anchors_t if anchors_t.dim() == 3 else anchors_t[None].expand(B, A, 5)
    if gt_t.shape[0]:
        tgt = ops.bbox2delta_rotated(anc.reshape(-1, 5).contiguous(), gt_t[gidx.view(-1)].contiguous(), means, stds).view(B, A, 5)
    else:
        tgt = torch.zeros((B, A, 5), device=gi.device)
    bt = torch.where(pos[..., None], tgt, torch.zeros_like(tgt))
    bw = pos[..., None].float().expand(B, A, 5)
    lw = neg.float() + pos.float()
    return gi, mo, lab, lw, bt, bw, pos.sum(1).clamp(min=1).sum().float(), neg.sum(1).clamp(min=1).sum().float()


def _check_equal(out, want, two_tier=False):
    """Every DECISION (gt_inds, labels, weights, encoded targets, counts) bit for bit; max_overlaps bit for bit in the
    exact mode and within the two-tier budget (rsdet_geom_fast.h kFastBudget = 2e-5; the contract is 1e-4) otherwise."""
    gi, mo, lab, lw, bt, bw, npos, nneg = want
    assert torch.equal(out["gt_inds"], gi), int((out["gt_inds"] != gi).sum())
    assert torch.equal(out["labels"], lab)
    assert torch.equal(out["label_weights"], lw)
    assert torch.equal(out["bbox_weights"], bw.contiguous())
    assert torch.equal(out["bbox_targets"], bt), float((out["bbox_targets"] - bt).abs().max())
    mo_ok = torch.where(gi >= 0, mo, out["max_overlaps"])      # ignored (invalid) anchors: -1 either way
    if two_tier:
        d = (out["max_overlaps"] - mo_ok).abs()
        assert float(d.max()) <= 2e-5 and ((out["max_overlaps"] == 0) == (mo_ok == 0)).all()
    else:
        assert torch.equal(out["max_overlaps"], mo_ok)
    assert float(out["totals"][0]) == float(npos) and float(out["totals"][1]) == float(nneg)


@pytest.mark.parametrize("two_tier", [False, True])
@pytest.mark.parametrize("ks,per_image", [([16, 100, 400, 40], False), ([16, 100, 400, 40], True), ([1], False),
                                          ([3, 0, 250], True), ([0, 0], False)])
def test_fused_anchor_target_equals_dense_chain(cuda, oracle_c, ks, per_image, two_tier):
    from rs_detection_amd import ops
    rng = np.random.default_rng(sum(ks) * 3 + per_image)
    anchors = s2anet_anchors()
    gts = np.concatenate([dota_boxes(rng, k) for k in ks] + [np.zeros((0, 5), np.float32)])
    labs = rng.integers(1, 16, sum(ks)).astype(np.int32)
    an = _refined(rng, anchors, len(ks)) if per_image else anchors
    at, gt, lt, ro = (torch.from_numpy(an).to(cuda), torch.from_numpy(gts).to(cuda), torch.from_numpy(labs).to(cuda),
                      _ro(ks, cuda))
    means, stds = (0.01, -0.02, 0.0, 0.03, 0.0), (1.0, 0.5, 2.0, 1.0, 0.25)
    for rep in range(4):          # the state buffer must come back zeroed: repeated calls give the same answer
        prep = ops.prepare_boxes(at, heavy_from=(None, 20480, 16384, 0)[rep])   # whole tiles / 4-row sub-tiles
        out = ops.anchor_target_rotated(at, gt, lt, ro, ks, 0.5, 0.4, 0.0, target_means=means, target_stds=stds,
                                        want_gt_inds=True, prepared=prep, two_tier=two_tier,
                                        prepared_gt=ops.prepare_boxes(gt) if rep % 2 else None)
        _check_equal(out, _dense_chain(ops, at, gt, lt, ro, ks, means=means, stds=stds), two_tier)
    # ... and the assignment equals the oracle's (restatement of assigner.py:111-170 over the reference-pinned IoU)
    g = int(np.argmax(ks))
    if ks[g]:
        r0 = int(np.sum(ks[:g]))
        ov = oracle_c.box_iou_rotated(gts[r0:r0 + ks[g]], np.ascontiguousarray(an[g] if per_image else an), 0)
        wgi, _, wl = oracle_c.assign_wrt_overlaps(ov, 0.5, 0.4, 0.0, True, True, labs[r0:r0 + ks[g]], 0)
        assert (out["gt_inds"][g].cpu().numpy() == wgi).all() and (out["labels"][g].cpu().numpy() == wl).all()


@pytest.mark.parametrize("two_tier", [False, True])
def test_fused_anchor_target_special_rows_ties_and_valid_mask(cuda, oracle_c, two_tier):
    """(i) a gt that overlaps NO anchor has row maximum 0 and, with min_pos_iou = 0, claims every anchor whose IoU with
    it is 0 -- i.e. all of them, unless a later gt overrides (assigner.py:151-160); (ii) exact IoU ties between
    anchors of a regular grid (gt centred between two cells): every tied anchor is assigned; (iii) a valid mask."""
    from rs_detection_amd import ops
    anchors = s2anet_anchors()
    A = anchors.shape[0]
    rng = np.random.default_rng(2)
    far = np.array([[5000., 5000., 30., 10., 0.3]], np.float32)                       # outside every anchor
    tie = np.array([[8 * 10 + 3.5 + 4.0, 8 * 7 + 3.5, 24., 24., 0.0]], np.float32)    # midway between two stride-8 cells
    for gts, ks in ((np.concatenate([dota_boxes(rng, 5), far, dota_boxes(rng, 3)]), [9]),
                    (np.concatenate([far, dota_boxes(rng, 4)]), [5]),
                    (np.concatenate([dota_boxes(rng, 4), far]), [5]),                  # last gt claims everything
                    (np.concatenate([tie, dota_boxes(rng, 6)]), [7]),
                    (np.concatenate([dota_boxes(rng, 20), far, tie]), [10, 12])):
        labs = rng.integers(1, 16, len(gts)).astype(np.int32)
        at, gt, lt, ro = (torch.from_numpy(anchors).to(cuda), torch.from_numpy(gts).to(cuda),
                          torch.from_numpy(labs).to(cuda), _ro(ks, cuda))
        out = ops.anchor_target_rotated(at, gt, lt, ro, ks, 0.5, 0.4, 0.0, want_gt_inds=True, two_tier=two_tier)
        _check_equal(out, _dense_chain(ops, at, gt, lt, ro, ks), two_tier)
        r0 = 0
        for g, k in enumerate(ks):
            ov = oracle_c.box_iou_rotated(gts[r0:r0 + k], anchors, 0)
            wgi, _, _ = oracle_c.assign_wrt_overlaps(ov, 0.5, 0.4, 0.0, True, True, labs[r0:r0 + k], 0)
            assert (out["gt_inds"][g].cpu().numpy() == wgi).all()
            r0 += k
    assert int((out["gt_inds"][1] == 12).sum()) >= 1            # the tie gt (last row of image 1) overrode the far gt
    # (iii) valid mask: the right-hand third of every level is outside the padded image
    ks = [30, 12]
    gts = np.concatenate([dota_boxes(rng, k) for k in ks])
    labs = rng.integers(1, 16, sum(ks)).astype(np.int32)
    valid = torch.from_numpy(np.stack([anchors[:, 0] < 700, anchors[:, 1] < 650])).to(cuda)
    at, gt, lt, ro = (torch.from_numpy(anchors).to(cuda), torch.from_numpy(gts).to(cuda),
                      torch.from_numpy(labs).to(cuda), _ro(ks, cuda))
    out = ops.anchor_target_rotated(at, gt, lt, ro, ks, 0.5, 0.4, 0.0, valid=valid, want_gt_inds=True, two_tier=two_tier)
    _check_equal(out, _dense_chain(ops, at, gt, lt, ro, ks, valid=valid), two_tier)
    assert (out["gt_inds"][~valid] == -1).all() and (out["label_weights"][~valid] == 0).all()


def test_prepared_cache_and_tile_table(cuda):
    from rs_detection_amd import ops
    from rs_detection_amd.ops import anchor_target as at
    a = torch.from_numpy(s2anet_anchors()).to(cuda)
    p1 = ops.prepare_boxes(a, cache=True)
    assert p1.heavy_from == 16384 + 4096                        # levels 2..4 (anchors of 128 px and up) are "large"
    assert at.heavy_from_boxes(a[:100]) == 100                  # a set without a large suffix: no split
    assert ops.prepare_boxes(a, cache=True) is p1               # same tensor, same version: no second sincos pass
    a.add_(0.0)                                                  # in-place write bumps the version counter
    assert ops.prepare_boxes(a, cache=True) is not p1
    table, t0, n = ops.row_tile_table([16, 100, 400, 40], cuda)
    assert n == 1 + 7 + 25 + 3 and t0.tolist() == [0, 1, 8, 33, 36]
    t = table.cpu().numpy()
    assert (t[0] == [0, 0, 16, 0]).all() and (t[7] == [1, 16 + 96, 4, 16]).all() and (t[35] == [3, 516 + 32, 8, 516]).all()
    assert ops.row_tile_table([16, 100, 400, 40], cuda)[0] is table


def test_split_dense_iou_queue_overflow(cuda):
    """More surviving pairs than the global queue holds (4 Mi): the tiles whose survivors did not fit are clipped by the
    fill phase of the second launch -- same values, nothing lost, and the shard counters come back to zero."""
    from rs_detection_amd import ops
    rng = np.random.default_rng(4)
    n = 2112
    base = np.array([100.0, 100.0, 60.0, 30.0, 0.3], np.float32)
    b1 = (base + rng.normal(0, [4, 4, 3, 2, 0.2], (n, 5))).astype(np.float32)      # one pile: every pair overlaps
    b2 = (base + rng.normal(0, [4, 4, 3, 2, 0.2], (n, 5))).astype(np.float32)
    t1, t2 = torch.from_numpy(b1).to(cuda), torch.from_numpy(b2).to(cuda)
    want = ops.box_iou_rotated(t1, t2)
    assert int((want > 0).sum()) > (4 << 20)
    for _ in range(2):                                                             # twice: the state is clean again
        got = ops.box_iou_rotated_tiled(t1, t2, split=True)
        assert torch.equal(got, want)


def test_two_tier_anchor_target_decisions_under_pressure(cuda, oracle_c):
    """The two-tier path where its second tier matters: (i) duplicated gts (every column they top has two candidates
    with EQUAL values: first-argmax must win), (ii) gts shifted by 1e-4 px copies (candidates within the budget but not
    equal), (iii) IoUs pinned next to the thresholds: anchors' own boxes scaled so that IoU = 0.5 / 0.4 +- 1e-6 ...,
    (iv) integer axis-aligned gts (the reference's fragile zone), (v) a pile: > 2 048 survivors in one tile.
    gt_inds / labels / targets must equal the dense chain (bit-exact reference-order values) AND the oracle."""
    from rs_detection_amd import ops
    rng = np.random.default_rng(11)
    anchors = s2anet_anchors()
    A = anchors.shape[0]
    base = dota_boxes(rng, 40)
    cases = []
    cases.append((np.concatenate([base, base[::-1].copy()]), [80]))                                   # (i)
    sh = base.copy()
    sh[:, :2] += rng.normal(0, 1e-4, (40, 2)).astype(np.float32)
    cases.append((np.concatenate([base, sh]), [40, 40]))                                              # (ii) two images
    cases.append((np.concatenate([base, sh, base]), [120]))                                           # (ii) one image
    # (iii) a gt concentric with an anchor, same angle, scaled: IoU = (s^2 if s < 1) -> s = sqrt(thr +- d)
    pick = anchors[rng.integers(0, A, 60)]
    g3 = pick.copy()
    sc = np.sqrt(np.concatenate([0.5 + rng.uniform(-3e-6, 3e-6, 30), 0.4 + rng.uniform(-3e-6, 3e-6, 30)])).astype(np.float32)
    g3[:, 2:4] *= sc[:, None]
    cases.append((g3, [60]))
    ib = dota_boxes(rng, 50)
    ib[:, :4] = np.round(ib[:, :4])
    ib[:, 4] = rng.choice([0, np.pi / 2, -np.pi / 2], 50)
    cases.append((ib.astype(np.float32), [50]))                                                       # (iv)
    pile = (np.array([512, 512, 300, 200, 0.3], np.float32) + rng.normal(0, [3, 3, 2, 2, 0.1], (300, 5))).astype(np.float32)
    cases.append((pile, [300]))                                                                       # (v)
    for gts, ks in cases:
        labs = rng.integers(1, 16, len(gts)).astype(np.int32)
        at, gt, lt, ro = (torch.from_numpy(anchors).to(cuda), torch.from_numpy(gts).to(cuda),
                          torch.from_numpy(labs).to(cuda), _ro(ks, cuda))
        for hf in (None, 20480):
            out = ops.anchor_target_rotated(at, gt, lt, ro, ks, 0.5, 0.4, 0.0, want_gt_inds=True, two_tier=True,
                                            prepared=ops.prepare_boxes(at, heavy_from=hf))
            _check_equal(out, _dense_chain(ops, at, gt, lt, ro, ks), True)
        r0 = 0
        for g, k in enumerate(ks):
            ov = oracle_c.box_iou_rotated(gts[r0:r0 + k], anchors, 0)
            wgi, _, _ = oracle_c.assign_wrt_overlaps(ov, 0.5, 0.4, 0.0, True, True, labs[r0:r0 + k], 0)
            assert (out["gt_inds"][g].cpu().numpy() == wgi).all()
            r0 += k


def test_two_tier_anchor_target_uses_the_fast_tier(cuda):
    """Guard against a silent all-exact path: at the step shape most max_overlaps must DIFFER in the last bits from the
    exact mode (they are Green-integral values), while every decision is identical."""
    from rs_detection_amd import ops
    rng = np.random.default_rng(3)
    ks = [16, 100, 400, 40]
    anchors = s2anet_anchors()
    gts = np.concatenate([dota_boxes(rng, k) for k in ks])
    at, gt, ro = torch.from_numpy(anchors).to(cuda), torch.from_numpy(gts).to(cuda), _ro(ks, cuda)
    a = ops.anchor_target_rotated(at, gt, None, ro, ks, 0.5, 0.4, 0.0, want_gt_inds=True, two_tier=True)
    b = ops.anchor_target_rotated(at, gt, None, ro, ks, 0.5, 0.4, 0.0, want_gt_inds=True, two_tier=False)
    assert torch.equal(a["gt_inds"], b["gt_inds"]) and torch.equal(a["bbox_targets"], b["bbox_targets"])
    nz = b["max_overlaps"] > 0
    frac_same = float((a["max_overlaps"][nz] == b["max_overlaps"][nz]).float().mean())
    assert frac_same < 0.6, frac_same
    assert float((a["max_overlaps"] - b["max_overlaps"]).abs().max()) <= 2e-5
