"""multi_apply / unmap / parse_losses with the reference semantics
(/root/reference/python/jdet/utils/general.py:50-79) on torch tensors."""
from functools import partial

import torch


def multi_apply(func, *args, **kwargs):
    pfunc = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


def unmap(data, count, inds, fill=0):
    """Scatter a subset (selected by bool mask ``inds``) back into ``count`` rows."""
    if data.dim() == 1:
        ret = torch.full((count,), fill, dtype=data.dtype, device=data.device)
        ret[inds] = data
    else:
        ret = torch.full((count,) + tuple(data.shape[1:]), fill, dtype=data.dtype, device=data.device)
        ret[inds, :] = data
    return ret


def parse_losses(losses):
    out = {}
    for name, value in losses.items():
        if isinstance(value, torch.Tensor):
            out[name] = value if value.dim() == 0 else value.mean()      # (the mean of a scalar is the scalar: no launch)
        elif isinstance(value, list):
            if value and all(isinstance(v, torch.Tensor) and v.dim() == 0 for v in value):
                # per-level scalars (the S2ANet / RetinaNet heads): one stack + one sum instead of a mean and an add per
                # level -- 2 launches each way instead of ~20 for a five-level loss
                out[name] = torch.stack(value).sum()
            else:
                out[name] = sum(v.mean() for v in value)
        else:
            raise TypeError('{} is not a tensor or list of tensors'.format(name))
    terms = [v for k, v in out.items() if 'loss' in k]
    if len(terms) > 2 and all(t.dim() == 0 and t.dtype == terms[0].dtype and t.device == terms[0].device for t in terms):
        total = torch.stack(terms).sum()                                 # two launches each way instead of one per term
    else:
        total = sum(terms)
    return total, out
