"""JDet config loader: ``.py`` / ``.yaml`` file -> attribute dict.

Behavioural mirror of /root/reference/python/jdet/config/config.py:16-165:
  * ``_base_`` (str | list) inheritance relative to the including file (:61-76);
  * ``_cover_: True`` replaces instead of merging, at any depth incl. the root (:79-101);
  * attribute access; a MISSING key reads as ``None`` (:24-27, Runner relies on it);
  * ``name`` / ``work_dir`` filled from the file name when absent (:107-110);
  * module objects imported by a ``.py`` config are dropped (:112-123).
"""
import copy
import inspect
import os
import runpy
from collections import OrderedDict

import yaml

__all__ = ["get_cfg", "init_cfg", "save_cfg", "print_cfg", "update_cfg", "Config"]

BASE_KEY = "_base_"
COVER_KEY = "_cover_"


def _strip_cover(node):
    if not isinstance(node, dict):
        return node
    return {k: _strip_cover(v) for k, v in node.items() if k != COVER_KEY}


def _merge(dst, src):
    """Merge ``src`` into ``dst`` in place with the reference's _cover_ rules."""
    assert isinstance(dst, dict) and isinstance(src, dict)
    if COVER_KEY in src:
        dst.clear()
        dst.update(_strip_cover(copy.deepcopy(src)))
        return
    for k, v in src.items():
        replace = (k not in dst or not isinstance(v, dict) or not isinstance(dst[k], dict)
                   or v.get(COVER_KEY, False))
        if replace:
            dst[k] = _strip_cover(copy.deepcopy(v))
        else:
            _merge(dst[k], v)


def _read_one(filename):
    ext = os.path.splitext(filename)[1]
    if not os.path.isfile(filename):
        raise FileNotFoundError(filename)
    if ext == ".yaml":
        with open(filename) as f:
            return yaml.safe_load(f.read()) or {}
    if ext == ".py":
        # executed in its own namespace (the reference import_module's it and then
        # deletes it from sys.modules: same visible effect, no sys.path games)
        ns = runpy.run_path(filename)
        return {k: v for k, v in ns.items() if not k.startswith('__')}
    raise AssertionError("unsupported config type.")


def _read_with_base(filename):
    cfg = _read_one(filename)
    if BASE_KEY not in cfg:
        return cfg
    bases = cfg.pop(BASE_KEY)
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    here = os.path.dirname(filename)
    for b in bases:
        _merge(merged, _read_with_base(os.path.join(here, b)))
    _merge(merged, cfg)
    return merged


class Config(OrderedDict):
    def __init__(self, *args):
        super().__init__()
        if len(args) == 1:
            self.load_from_file(args[0])
        else:
            assert len(args) == 0

    def __getattr__(self, name):
        return self[name] if name in self else None

    def __setattr__(self, name, value):
        self[name] = value

    def load_from_file(self, filename):
        raw = _read_with_base(filename)
        self.clear()
        self.update(self._wrap(raw))
        if self.name is None:
            self.name = os.path.splitext(os.path.basename(filename))[0]
        if self.work_dir is None:
            self.work_dir = f"work_dirs/{self.name}"

    @classmethod
    def _wrap(cls, node):
        if isinstance(node, dict):
            out = cls()
            for k, v in node.items():
                if not inspect.ismodule(v):
                    out[k] = cls._wrap(v)
            return out
        if isinstance(node, list):
            return [cls._wrap(v) for v in node if not inspect.ismodule(v)]
        return copy.deepcopy(node)

    def dump(self):
        out = {}
        for k, v in self.items():
            if isinstance(v, Config):
                v = v.dump()
            if isinstance(v, list):
                v = [x.dump() if isinstance(x, Config) else x for x in v]
            out[k] = v
        return out


_cfg = Config()


def init_cfg(filename):
    print("Loading config from: ", filename)
    _cfg.load_from_file(filename)


def get_cfg():
    return _cfg


def update_cfg(args):
    _cfg.update(args)


def save_cfg(save_file):
    with open(save_file, "w") as f:
        f.write(yaml.dump(_cfg.dump()))


def print_cfg():
    print(yaml.dump(_cfg.dump()))
