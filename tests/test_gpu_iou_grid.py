"""GPU: the grid-aware dense rotated IoU (csrc/iou_grid.hip, ops.box_iou_rotated_grid) against the tile form of the
same two-tier op (csrc/iou_fast.hip, ops.box_iou_rotated_fast): BIT-IDENTICAL matrices -- both run the same tests and
the same two clippers on every pair that can overlap; the grid form only finds those pairs in closed form (cell windows
per pyramid level) instead of testing all of them.  tests/test_gpu_iou_fast.py pins the tile form to the reference's CPU
source; one direct comparison with it is repeated here."""
import numpy as np
import pytest
import torch

from conftest import dota_boxes

pytestmark = pytest.mark.gpu


def _same(a, b):
    return bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())


def _spec(img, strides=(8, 16, 32, 64, 128)):
    from rs_detection_amd.ops.anchor_target import s2anet_grid_spec
    import math
    return s2anet_grid_spec([(math.ceil(img / s), math.ceil(img / s)) for s in strides], strides)


def _gts(rng, ks, span):
    out = []
    for k in ks:
        out.append(dota_boxes(rng, k, span))
    return np.concatenate(out)


@pytest.mark.parametrize("version", [0, 1])
@pytest.mark.parametrize("img,ks", [(1024, [16, 100, 400, 40]), (1000, [33, 7]), (256, [12, 0, 5])])
def test_grid_iou_equals_the_tile_form_on_generated_anchors(cuda, img, ks, version):
    from rs_detection_amd import ops
    from rs_detection_amd.utils import synthetic as syn
    rng = np.random.default_rng(img + version)
    anchors = torch.from_numpy(syn.s2anet_anchor_grid(img)).to(cuda)
    gt = torch.from_numpy(_gts(rng, ks, img)).to(cuda)
    ro = torch.tensor(np.concatenate([[0], np.cumsum(ks)]), dtype=torch.int32, device=cuda)
    want = ops.box_iou_rotated_fast(gt, anchors, ro, ks=ks, version=version)
    got = ops.box_iou_rotated_grid(gt, anchors, _spec(img), version=version)
    assert got.shape == want.shape and _same(got, want), int((got != want).sum())
    assert int((want > 0).sum()) > 10 * sum(ks)                      # the comparison saw overlapping pairs


def test_grid_iou_special_rows(cuda):
    """Rows the windows must not lose: NaN / Inf boxes (every cell a candidate: NaN in, NaN out as in the tile form),
    boxes far outside the image, larger than the image, of zero / negative size, integer axis-aligned boxes on the anchor
    lattice (the reference's fragile zone: most pairs go through tier 2), a box whose window covers a whole level."""
    from rs_detection_amd import ops
    from rs_detection_amd.utils import synthetic as syn
    anchors = torch.from_numpy(syn.s2anet_anchor_grid(1024)).to(cuda)
    rng = np.random.default_rng(5)
    rows = [[np.nan, 100, 30, 20, 0.3], [100, 100, np.inf, 20, 0.3], [100, 100, 30, 20, np.nan], [1e6, 5e5, 80, 40, 0.2],
            [-300, -250, 90, 50, 1.0], [512, 512, 3000, 2000, 0.7], [300, 300, 0, 0, 0], [300, 300, -40, 20, 0.5],
            [3.5, 3.5, 32, 32, 0], [11.5, 3.5, 32, 32, np.pi / 2], [63.5, 63.5, 512, 512, 0], [512, 512, 1024, 1024, 0],
            [1023.9, 1023.9, 10, 5, 2.0], [0, 0, 1e-3, 1e-3, 0.1], [3.5, 3.5, 32.00001, 32, 1e-7]]
    ib = np.stack([rng.integers(0, 128, 40) * 8 + 3.5, rng.integers(0, 128, 40) * 8 + 3.5, rng.integers(1, 9, 40) * 8,
                   rng.integers(1, 9, 40) * 8, rng.choice([0, np.pi / 2, -np.pi / 2], 40)], 1)
    gt = torch.from_numpy(np.concatenate([np.asarray(rows, np.float64), ib]).astype(np.float32)).to(cuda)
    for version in (0, 1):
        want = ops.box_iou_rotated_fast(gt, anchors, version=version)
        got = ops.box_iou_rotated_grid(gt, anchors, _spec(1024), version=version)
        bad = ~((got == want) | (torch.isnan(got) & torch.isnan(want)))
        assert not bool(bad.any()), (version, bad.nonzero()[:5].tolist())


def test_grid_spec_is_the_anchor_generator(cuda):
    """The closed form the kernel uses for a cell's box == the anchors the model generates (models/boxes/
    anchor_generator.py), for square and non-square pyramids."""
    from rs_detection_amd.models.boxes.anchor_generator import AnchorGeneratorRotatedS2ANet
    from rs_detection_amd.ops.anchor_target import s2anet_grid_spec
    strides, sizes = (8, 16, 32, 64, 128), [(100, 128), (50, 64), (25, 32), (13, 16), (7, 8)]
    gens = [AnchorGeneratorRotatedS2ANet(s, [4], [1.0]) for s in strides]
    want = torch.cat([g.grid_anchors(sz, s, device=cuda) for g, sz, s in zip(gens, sizes, strides)])
    spec = s2anet_grid_spec(sizes, strides)
    assert spec.n == want.shape[0] and bool((spec.boxes(cuda) == want).all()) and spec.matches(want)
    moved = want.clone()
    moved[5, 0] += 1e-3
    assert not spec.matches(moved)


def test_grid_iou_against_the_reference_cpu_source(cuda):
    import oracle
    from rs_detection_amd import ops
    from rs_detection_amd.utils import synthetic as syn
    r = oracle.ref()
    impl = r if r.available else oracle.c()
    rng = np.random.default_rng(3)
    a = syn.s2anet_anchor_grid(512)
    g = dota_boxes(rng, 60, 512)
    want = impl.box_iou_rotated(g, a, 0)
    got = ops.box_iou_rotated_grid(torch.from_numpy(g).to(cuda), torch.from_numpy(a).to(cuda), _spec(512)).cpu().numpy()
    assert np.abs(got - want).max() <= 2e-5 and ((want == 0) == (got == 0)).all()


def test_grid_spec_must_be_the_columns(cuda):
    """The entry point never reads the columns: the wrapper refuses a tensor that is not the grid of the spec."""
    from rs_detection_amd import ops, _lib
    from rs_detection_amd.utils import synthetic as syn
    anchors = torch.from_numpy(syn.s2anet_anchor_grid(256)).to(cuda)
    gt = torch.zeros((2, 5), device=cuda)
    with pytest.raises(_lib.RsdetError):
        ops.box_iou_rotated_grid(gt, anchors, _spec(512))
    with pytest.raises(_lib.RsdetError):
        ops.box_iou_rotated_grid(gt, torch.from_numpy(syn.refined_anchor_grid(img=256)).to(cuda), _spec(256))
