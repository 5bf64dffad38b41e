"""GPU: the two-tier dense rotated IoU (csrc/iou_fast.hip, ops.box_iou_rotated_fast) against the reference's own CPU
source (oracle/_ref, or the oracle restatement when the reference was not mounted at build time).

Contract (BASELINE north_star): |IoU - reference| <= 1e-4.  Asserted here: <= 2e-5 (the decision budget the two-tier
callers use, rsdet_geom_fast.h kFastBudget) on >= 1.2e7 overlapping pairs of ten box families, exact zeros wherever the
reference returns zero, and the reference's own fragile cases (identical boxes, integer axis-aligned boxes, corners on
edges: its hull scan drops vertices there) reproduced through tier 2."""
import numpy as np
import pytest
import torch

import oracle
from conftest import dota_boxes, degenerate_boxes, s2anet_anchors

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _ref(b1, b2, v=0):
    r = oracle.ref()
    return (r if r.available else oracle.c()).box_iou_rotated(np.ascontiguousarray(b1), np.ascontiguousarray(b2), v)


def _fast(cuda, b1, b2, v=0, **kw):
    from rs_detection_amd import ops
    return ops.box_iou_rotated_fast(torch.from_numpy(b1).to(cuda), torch.from_numpy(b2).to(cuda), version=v,
                                    **kw).cpu().numpy()


def _check(got, want, name):
    d = np.abs(got - want)
    assert np.nanmax(d) <= TOL, (name, float(np.nanmax(d)))
    assert ((want == 0) == (got == 0)).all(), (name, int(((want == 0) != (got == 0)).sum()))   # exact zeros, both ways
    return int((want > 0).sum())


def _families(rng):
    yield "dota clustered", dota_boxes(rng, 1500, 300), dota_boxes(rng, 2500, 300)
    b = dota_boxes(rng, 800, 200)
    b[:, 3] = rng.uniform(0.01, 2, 800)
    yield "thin", b, dota_boxes(rng, 1500, 200)
    b = dota_boxes(rng, 800, 200)
    b[:, 2:4] *= 20
    yield "huge vs normal", b, dota_boxes(rng, 1500, 200)
    base = dota_boxes(rng, 1200, 200)
    for eps in (0, 1e-5, 1e-3, 1e-2, 0.1):
        yield "jitter %g" % eps, base, (base + rng.normal(0, eps, base.shape)).astype(np.float32)
    ib = np.stack([rng.integers(0, 40, 1200), rng.integers(0, 40, 1200), rng.integers(1, 20, 1200) * 2,
                   rng.integers(1, 20, 1200) * 2, rng.choice([0, np.pi / 2, np.pi, -np.pi / 2], 1200)], 1).astype(np.float32)
    yield "integer axis-aligned", ib[:600], ib[600:]
    ib2 = ib.copy()
    ang = np.arctan2(3, 4)
    ib2[:, 4] = rng.choice([ang, -ang, ang + np.pi / 2, 0], 1200)
    yield "integer 3-4-5", ib2[:600], ib2[600:]
    yield "integer 3-4-5 vs axis", ib2[:600], ib[600:]
    pb = dota_boxes(rng, 1200, 150)
    pb[:, 4] = rng.choice([0.3, 0.3 + np.pi / 2], 1200)
    yield "parallel mod pi/2", pb[:600], pb[600:]
    g = dota_boxes(rng, 600, 1024)
    g[:, :4] = np.round(g[:, :4] * 2) / 2
    g[:, 4] = rng.choice([0, -np.pi / 2, np.pi / 2], 600)
    yield "half-integer axis gts vs anchors", g, s2anet_anchors()[-1364 - 1024:]


@pytest.mark.parametrize("version", [0, 1])
def test_fast_iou_fuzz_against_the_reference(cuda, version):
    rng = np.random.default_rng(100 + version)
    n_over = 0
    for rep in range(2):
        for name, b1, b2 in _families(rng):
            n_over += _check(_fast(cuda, b1, b2, version), _ref(b1, b2, version), name)
    assert n_over >= 6_000_000, n_over            # x 2 versions: >= 1.2e7 overlapping pairs


def test_fast_iou_degenerate_boxes_take_the_reference_path(cuda):
    deg = degenerate_boxes()
    for v in (0, 1):
        got, want = _fast(cuda, deg, deg, v), _ref(deg, deg, v)
        fin = np.isfinite(want)
        assert np.abs(got - want)[fin].max() <= TOL
        assert (np.isnan(got) == np.isnan(want)).all()


def test_fast_iou_step_shape_grouped_and_ragged(cuda):
    """The S2ANet step shape (4 images, K = 16/100/400/40, shared 21 824-anchor grid) through the tile table, without it,
    with per-image column sets, with n2 % 4 != 0 (scalar zero fill) and a ragged last tile."""
    from rs_detection_amd import ops
    rng = np.random.default_rng(7)
    ks = [16, 100, 400, 40]
    gt = np.concatenate([dota_boxes(rng, k) for k in ks])
    ro = torch.tensor(np.concatenate([[0], np.cumsum(ks)]), dtype=torch.int32, device=cuda)
    A = s2anet_anchors()
    want = np.zeros((sum(ks), len(A)), np.float32)
    r0 = 0
    for k in ks:
        want[r0:r0 + k] = _ref(gt[r0:r0 + k], A)
        r0 += k
    g, a = torch.from_numpy(gt).to(cuda), torch.from_numpy(A).to(cuda)
    for kw in (dict(ks=ks), dict(max_rows=max(ks))):
        got = ops.box_iou_rotated_fast(g, a, ro, **kw).cpu().numpy()
        # rows of image i only meet image i's... the anchor set is shared: every row against all anchors
        assert np.abs(got - want).max() <= TOL and ((got == 0) == (want == 0)).all()
    # per-image column sets (G, A', 5) with A' % 4 != 0 and a ragged last tile
    Ap = 1000 + 3
    cols = np.stack([dota_boxes(rng, Ap, 1024, 16, 300, 200) for _ in ks])
    got = ops.box_iou_rotated_fast(g, torch.from_numpy(cols).to(cuda), ro, ks=ks).cpu().numpy()
    r0 = 0
    for i, k in enumerate(ks):
        w = _ref(gt[r0:r0 + k], cols[i])
        assert np.abs(got[r0:r0 + k] - w).max() <= TOL and ((got[r0:r0 + k] == 0) == (w == 0)).all()
        r0 += k
    # plain (ungrouped) call, empty sides
    got = ops.box_iou_rotated_fast(g[:50], a[:777]).cpu().numpy()
    assert np.abs(got - _ref(gt[:50], A[:777])).max() <= TOL
    assert ops.box_iou_rotated_fast(g[:0], a).shape == (0, len(A))


def test_fast_iou_dense_tile_overflows_the_flag_list(cuda):
    """A tile whose every pair is flagged (identical integer boxes: 32 x 256 = 8 192 > the 1 024-entry list): the whole
    tile goes through tier 2 and equals the reference."""
    b = np.tile(np.array([[50, 50, 20, 10, 0]], np.float32), (300, 1))
    got, want = _fast(cuda, b[:40], b), _ref(b[:40], b)
    assert np.abs(got - want).max() <= TOL
