"""Loader-only rate (tiles/s) of the DOTA training pipeline on a generated 1024 x 1024 PNG set: PIL decode +
RotatedResize + flips + RandomRotateAug + Pad + Normalize + collate (batch 4) [+ pinned H2D on a side stream when a GPU
is there], for num_workers in argv (default 0 4 8 16).  Usage: python profiles/scripts/loader_rate.py [n_images] [workers...]"""
import os
import pickle
import sys
import tempfile
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rs_detection_amd.data import DOTADataset  # noqa: E402
from rs_detection_amd.data.synthetic import render_tile  # noqa: E402
from rs_detection_amd.data.loader import prefetch_to_device  # noqa: E402

TF = [dict(type="RotatedResize", min_size=1024, max_size=1024), dict(type="RotatedRandomFlip", prob=0.5, direction="horizontal"),
      dict(type="RotatedRandomFlip", prob=0.5, direction="vertical"), dict(type="RandomRotateAug", random_rotate_on=True),
      dict(type="Pad", size_divisor=32),
      dict(type="Normalize", mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_bgr=False)]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    workers = [int(w) for w in sys.argv[2:]] or [0, 4, 8, 16]
    root = tempfile.mkdtemp(prefix="rsdet_loader_")
    os.makedirs(os.path.join(root, "images"))
    rng = np.random.default_rng(0)
    infos = []
    for i in range(n):
        k = (16, 100, 400, 40)[i % 4]
        w = rng.uniform(10, 160, k)
        b = np.stack([rng.uniform(0, 1024, k), rng.uniform(0, 1024, k), w, rng.uniform(5, np.minimum(w, 64)),
                      rng.uniform(-np.pi / 4, 3 * np.pi / 4, k)], 1).astype(np.float32)
        lab = rng.integers(1, 16, k).astype(np.int32)
        img = render_tile(b, lab, 1024, rng)                              # (3, H, W) float in [0, 1]-ish
        img8 = np.clip(np.transpose(img, (1, 2, 0)) * 255, 0, 255).astype(np.uint8)
        Image.fromarray(img8).save(os.path.join(root, "images", "P%04d.png" % i))
        infos.append(dict(filename="P%04d.png" % i, width=1024, height=1024,
                          ann=dict(bboxes=b, labels=lab, bboxes_ignore=np.zeros((0, 5), np.float32))))
    with open(os.path.join(root, "labels.pkl"), "wb") as f:
        pickle.dump(infos, f)
    import torch
    dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    print("images", n, "png bytes/img %.0f KB" % (os.path.getsize(os.path.join(root, "images", "P0000.png")) / 1024),
          "host cores", os.cpu_count(), "device", dev)
    for nw in workers:
        ds = DOTADataset(dataset_dir=root, transforms=TF, batch_size=4, shuffle=True, seed=1, num_workers=nw)
        try:
            ds.set_epoch(0)
            for _ in prefetch_to_device(ds, dev):         # epoch 0: starts the workers, warms the page cache
                pass
            t0, tiles = time.perf_counter(), 0
            for ep in (1, 2):
                ds.set_epoch(ep)
                for images, _ in prefetch_to_device(ds, dev):
                    tiles += images.shape[0]
            if dev.type == "cuda":
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("num_workers %2d: %.1f tiles/s (%d tiles in %.2f s)" % (nw, tiles / dt, tiles, dt), flush=True)
        finally:
            if getattr(ds, "_worker_pool", None) is not None:
                ds._worker_pool.close()


if __name__ == "__main__":
    main()
