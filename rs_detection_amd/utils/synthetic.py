"""Deterministic synthetic DOTA-shaped inputs: tiles, targets, micro-bench boxes."""
import math

import numpy as np

K_CYCLE = (16, 100, 400, 40)


def dota_gt_boxes(rng, k, span=1024.0):
    """cx,cy~U[0,span); long side w~U[10,160); h~U[5,min(w,64)); theta~U[-pi/4,3pi/4) (le135)."""
    w = rng.uniform(10, 160, k)
    h = rng.uniform(5, np.minimum(w, 64))
    return np.stack([rng.uniform(0, span, k), rng.uniform(0, span, k), w, h,
                     rng.uniform(-math.pi / 4, 3 * math.pi / 4, k)], 1).astype(np.float32)


def s2anet_anchor_grid(img=1024, strides=(8, 16, 32, 64, 128), scale=4):
    """models/boxes/anchor_generator.py:22-78: one square anchor per cell, x fastest."""
    out = []
    for s in strides:
        f = int(math.ceil(img / s))
        xs = np.arange(f, dtype=np.float32) * s + 0.5 * (s - 1)
        a = np.zeros((f * f, 5), np.float32)
        a[:, 0] = np.tile(xs, f)
        a[:, 1] = np.repeat(xs, f)
        a[:, 2] = a[:, 3] = scale * s
        out.append(a)
    return np.concatenate(out)


def refined_anchor_grid(seed=7, **kw):
    """grid perturbed by dxy~N(0,4px), dlog(w,h)~N(0,.2), dtheta~N(0,.3)."""
    rng = np.random.default_rng(seed)
    a = s2anet_anchor_grid(**kw)
    n = a.shape[0]
    a[:, :2] += rng.normal(0, 4, (n, 2)).astype(np.float32)
    a[:, 2:4] *= np.exp(rng.normal(0, 0.2, (n, 2))).astype(np.float32)
    a[:, 4] += rng.normal(0, 0.3, n).astype(np.float32)
    return a


def nms_cluster_boxes(m, seed=11, n_centres=200):
    """200 cluster centres, members jittered (3 px, 0.1 rad), scores~U(.05,1), labels~U{0..14}."""
    rng = np.random.default_rng(seed)
    centres = dota_gt_boxes(rng, n_centres)
    idx = rng.integers(0, n_centres, m)
    d = centres[idx].copy()
    d[:, :2] += rng.normal(0, 3, (m, 2)).astype(np.float32)
    d[:, 4] += rng.normal(0, 0.1, m).astype(np.float32)
    scores = rng.uniform(0.05, 1, m).astype(np.float32)
    labels = rng.integers(0, 15, m).astype(np.int32)
    return d, scores, labels


def synthetic_targets(batch, rank=0, it=0, num_classes=15, img=1024, k_shift=0, ks=None):
    """Per-tile target dicts with the reference schema (data/custom.py:75-88).  ``k_shift`` rotates the K cycle (with a
    batch of len(K_CYCLE) tiles every ``it`` would otherwise put the same K in the same slot); ``ks`` gives the gt
    counts explicitly (the fresh-K leg of bench.py)."""
    rng = np.random.default_rng(1234 + 1000 * rank + it)
    out = []
    for b in range(batch):
        k = K_CYCLE[(it * batch + b + k_shift) % len(K_CYCLE)] if ks is None else int(ks[b])
        out.append(dict(rboxes=dota_gt_boxes(rng, k, img),
                        labels=rng.integers(1, num_classes + 1, k).astype(np.int32),
                        rboxes_ignore=np.zeros((0, 5), np.float32),
                        img_size=(img, img), pad_shape=(img, img), scale_factor=1.0, img_file="synthetic"))
    return out
