"""JDet's operator registry, kept verbatim in behaviour so existing configs load unchanged.

Mirror of /root/reference/python/jdet/utils/registry.py:1-63: ``@REG.register_module()``,
``build_from_cfg(cfg, REG, **kw)`` (dict -> pop ``type`` -> ctor kwargs; str -> no-arg
ctor; list -> Sequential; None -> None; TypeError re-raised with the class name) and
the same 14 registries.
"""


class Registry:
    def __init__(self):
        self._modules = {}

    def register_module(self, name=None, module=None):
        def _register(mod):
            key = mod.__name__ if name is None else name
            assert key not in self._modules, f"{key} is already registered."
            self._modules[key] = mod
            return mod

        return _register(module) if module is not None else _register

    def get(self, name):
        assert name in self._modules, f"{name} is not registered."
        return self._modules[name]

    def __contains__(self, name):
        return name in self._modules


def build_from_cfg(cfg, registry, **kwargs):
    if isinstance(cfg, str):
        return registry.get(cfg)(**kwargs)
    if isinstance(cfg, dict):
        args = dict(cfg)
        args.update(kwargs)
        obj_cls = registry.get(args.pop('type'))
        try:
            return obj_cls(**args)
        except TypeError as e:
            msg = str(e) if "<class" in str(e) else f"{obj_cls}.{e}"
            raise TypeError(msg)
    if isinstance(cfg, list):
        from torch import nn
        return nn.Sequential(*[build_from_cfg(c, registry, **kwargs) for c in cfg])
    if cfg is None:
        return None
    raise TypeError(f"type {type(cfg)} not support")


DATASETS = Registry()
TRANSFORMS = Registry()
MODELS = Registry()
BACKBONES = Registry()
HEADS = Registry()
LOSSES = Registry()
OPTIMS = Registry()
BRICKS = Registry()
NECKS = Registry()
SCHEDULERS = Registry()
BOXES = Registry()
HOOKS = Registry()
ROI_EXTRACTORS = Registry()
SHARED_HEADS = Registry()
