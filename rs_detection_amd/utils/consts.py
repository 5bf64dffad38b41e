"""Small constant tensors (coder means / stds, scalar fills, [0, K] row offsets) as cached DEVICE tensors.

``x.new_tensor([...])`` / ``torch.tensor([...], device=...)`` build the value on the host and copy it from pageable memory:
a synchronous transfer of ~150 us each time (26 per Oriented R-CNN step = 3.7 ms of host time, measured with
profiles/scripts/host_prof_orcnn.py).  The values never change, so each (values, dtype, device) is uploaded once."""
from collections import OrderedDict

import torch

_CACHE = OrderedDict()
_MAX = 256


def const_tensor(values, like=None, dtype=None, device=None):
    """The tensor ``like.new_tensor(values)`` would give (dtype / device from ``like`` unless given), cached.  Treat it as
    read-only."""
    dtype = dtype or (like.dtype if like is not None else torch.float32)
    device = device or (like.device if like is not None else torch.device("cpu"))
    vals = tuple(float(v) for v in values) if isinstance(values, (list, tuple)) else (float(values),)
    key = (vals, isinstance(values, (list, tuple)), dtype, str(device))
    t = _CACHE.get(key)
    if t is None:
        t = torch.tensor(list(vals) if key[1] else vals[0], dtype=dtype, device=device)
        _CACHE[key] = t
        if len(_CACHE) > _MAX:
            _CACHE.popitem(last=False)
    else:
        _CACHE.move_to_end(key)
    return t
