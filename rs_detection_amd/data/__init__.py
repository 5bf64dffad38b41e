"""Evaluation-side host logic: tile-result merging and DOTA mAP, with the polygon IoU / NMS
on the GPU (ops/nms_poly.py).  Datasets and transforms of the reference are out of scope."""
from . import transforms  # noqa: F401,E402  (registers TRANSFORMS)
from .custom import CustomDataset  # noqa: F401,E402
from .dota import DOTADataset  # noqa: F401,E402
from .synthetic import SyntheticDOTADataset  # noqa: F401,E402
from .fair import FAIR1M_1_5_Dataset, FAIRDataset  # noqa: F401,E402
from .image import ImageDataset  # noqa: F401,E402


def batch_to_device(images, targets, device):
    """NumPy batch of a dataset -> what the detectors take: images (N,3,H,W) float tensor on ``device`` and the same
    target dicts with their box / label arrays as tensors on ``device`` (everything else untouched)."""
    import numpy as np
    import torch
    out = []
    for t in targets:
        t = dict(t)
        for k in ("rboxes", "hboxes", "polys", "labels", "rboxes_ignore"):
            if isinstance(t.get(k), np.ndarray):
                t[k] = torch.from_numpy(np.ascontiguousarray(t[k])).to(device)
        out.append(t)
    return torch.from_numpy(np.ascontiguousarray(images)).to(device), out
