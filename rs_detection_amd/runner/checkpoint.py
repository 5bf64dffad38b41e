"""Checkpoints in the reference's on-disk form (/root/reference/python/jdet/runner/runner.py:251-290).

A JDet checkpoint ``ckpt_<epoch>.pkl`` is ``jt.save`` of
``{"meta": {...}, "model": state_dict, "scheduler": ..., "optimizer": ...}``; ``jt.save`` pickles the structure
with every ``jt.Var`` turned into a NumPy array, so such a file unpickles WITHOUT Jittor.  ``load`` also accepts
the two other layouts the reference's ``Runner.load`` accepts (a ``state_dict`` key, or a bare parameter dict --
e.g. ``jittorhub://resnet50.pkl``, :273-279).  Module names in this repo follow the reference's (resnet.py
conv1/bn1/layerN.M..., s2anet_head.py fam_reg_convs..., or_conv.weight, ...), so parameters map by name;
``load_parameters`` reports what did not match instead of raising, like Jittor's ``load_parameters``."""
import pickle
import time

import numpy as np
import torch


def _to_numpy(obj):
    if isinstance(obj, torch.Tensor):
        t = obj.detach().cpu()
        return (t.float() if t.dtype == torch.bfloat16 else t).numpy()   # NumPy has no bf16: widen (exact)
    if isinstance(obj, dict):
        return {k: _to_numpy(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_numpy(v) for v in obj)
    return obj


def save_checkpoint(path, model, optimizer=None, scheduler=None, meta=None):
    """``Runner.save`` (:251-269): same top-level keys, arrays as NumPy, plain pickle.  Where the optimizer keeps fp32
    masters of bf16 model weights (FusedSGD) the checkpoint stores the MASTERS under the model's names: a JDet-layout
    fp32 file either way."""
    state = dict(model.state_dict())
    if optimizer is not None and hasattr(optimizer, "master_state_dict"):
        state.update(optimizer.master_state_dict(model))
    data = {"meta": dict(meta or {}, save_time=time.strftime("%Y%m%d_%H%M%S")),
            "model": _to_numpy(state),
            "scheduler": _to_numpy(scheduler.parameters()) if scheduler is not None else {},
            "optimizer": _to_numpy(optimizer.state_dict()) if optimizer is not None else {}}
    with open(path, "wb") as f:
        pickle.dump(data, f, protocol=4)
    return data


def read_checkpoint(path):
    """Unpickle a JDet / Jittor ``.pkl`` (no Jittor needed: ``jt.save`` stores NumPy arrays)."""
    with open(path, "rb") as f:
        return pickle.load(f)


def model_parameters(data):
    """:273-279: ``model`` key, else ``state_dict`` key, else the dict itself."""
    if isinstance(data, dict) and "model" in data:
        return data["model"]
    if isinstance(data, dict) and "state_dict" in data:
        return data["state_dict"]
    return data


def load_parameters(model, params, verbose=False):
    """Copy ``params`` (name -> ndarray / tensor) into ``model`` by name.  Returns (loaded, missing, unexpected,
    mismatched); nothing is raised for the last three (Jittor's load_parameters warns and carries on)."""
    own = model.state_dict()
    loaded, unexpected, mismatched = [], [], []
    with torch.no_grad():
        for k, v in params.items():
            if k not in own:
                unexpected.append(k)
                continue
            t = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else torch.as_tensor(v)
            if tuple(t.shape) != tuple(own[k].shape):
                if t.numel() == own[k].numel() and own[k].dim() <= 1:   # e.g. a scalar counter saved as (1,)
                    t = t.reshape(own[k].shape)
                else:
                    mismatched.append((k, tuple(t.shape), tuple(own[k].shape)))
                    continue
            own[k].copy_(t.to(own[k].dtype))
            loaded.append(k)
    missing = [k for k in own if k not in params]
    if verbose:
        print("checkpoint: %d loaded, %d missing, %d unexpected, %d shape mismatches" %
              (len(loaded), len(missing), len(unexpected), len(mismatched)))
    return loaded, missing, unexpected, mismatched


def average_checkpoints(paths):
    """tools/get_SWA_model.py: element-wise mean of the ``model`` dicts of several checkpoints (SWA)."""
    acc, n = None, 0
    for p in paths:
        m = model_parameters(read_checkpoint(p))
        if acc is None:
            acc = {k: np.array(v, dtype=np.float64) if np.issubdtype(np.asarray(v).dtype, np.floating) else np.asarray(v)
                   for k, v in m.items()}
        else:
            for k, v in m.items():
                if np.issubdtype(np.asarray(v).dtype, np.floating):
                    acc[k] = acc[k] + np.asarray(v, dtype=np.float64)
        n += 1
    return {k: (v / n).astype(np.float32) if v.dtype == np.float64 else v for k, v in acc.items()}


def load_pretrained(model, name, pretrained):
    """``pretrained=...`` of a backbone factory (resnet.py:206-209 ``jittorhub://resnet50.pkl``, van.py ``load_model``).
    ``pretrained`` may be a path to a ``.pkl`` / ``.pth`` parameter dict; ``True`` looks for ``$RSDET_PRETRAINED_DIR/<name>.pkl``
    (there is no network to download from).  When nothing can be loaded the weights stay at random init and that is
    said LOUDLY -- an accuracy run on random backbones is not what the config asked for."""
    import os
    import warnings
    if not pretrained:
        return None
    path = pretrained if isinstance(pretrained, str) else None
    if path is None and os.environ.get("RSDET_PRETRAINED_DIR"):
        for ext in (".pkl", ".pth"):
            cand = os.path.join(os.environ["RSDET_PRETRAINED_DIR"], name + ext)
            if os.path.isfile(cand):
                path = cand
                break
    if path is None or not os.path.isfile(path):
        warnings.warn("pretrained=%r requested for %s but no weight file is available offline (set RSDET_PRETRAINED_DIR "
                      "or pass a path; Runner.load(<converted .pkl>, model_only=True) also works): the backbone stays at "
                      "RANDOM initialisation" % (pretrained, name), RuntimeWarning, stacklevel=3)
        return None
    data = torch.load(path, map_location="cpu") if path.endswith(".pth") else read_checkpoint(path)
    rep = load_parameters(model, model_parameters(data))
    if not rep[0]:
        warnings.warn("pretrained file %s matched no parameter of %s" % (path, name), RuntimeWarning, stacklevel=3)
    return rep
