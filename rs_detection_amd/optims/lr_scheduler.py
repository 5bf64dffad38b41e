"""Warm-up + step / cosine LR schedules (/root/reference/python/jdet/optims/lr_scheduler.py:8-60,196-236,274-320).
``step(iters, epochs, by_epoch=True)`` is called once per iteration by the Runner."""
import math

from rs_detection_amd.utils.registry import SCHEDULERS


@SCHEDULERS.register_module()
class WarmUpLR:
    def __init__(self, optimizer, warmup_ratio=1.0 / 3, warmup_iters=500, warmup=None):
        self.optimizer, self.warmup_ratio, self.warmup_iters, self.warmup = optimizer, warmup_ratio, warmup_iters, warmup
        self.base_lr = optimizer.lr
        self.base_lr_pg = [pg.get("lr", optimizer.lr) for pg in optimizer.param_groups]
        self.step(0, 0)

    def get_warmup_lr(self, lr, cur_iters):
        if self.warmup == 'constant':
            k = self.warmup_ratio
        elif self.warmup == 'linear':
            k = 1 - (1 - cur_iters / self.warmup_iters) * (1 - self.warmup_ratio)
        elif self.warmup == 'exp':
            k = self.warmup_ratio ** (1 - cur_iters / self.warmup_iters)
        else:
            raise ValueError(self.warmup)
        return k * lr

    def get_lr(self, lr, steps):
        return lr

    def _update_lr(self, steps, fn):
        self.optimizer.lr = fn(self.base_lr, steps)
        for i, pg in enumerate(self.optimizer.param_groups):
            pg["lr"] = fn(self.base_lr_pg[i], steps)

    def step(self, iters, epochs, by_epoch=True):
        if self.warmup is not None and iters < self.warmup_iters:
            self._update_lr(iters, self.get_warmup_lr)
        elif by_epoch:
            self._update_lr(epochs, self.get_lr)
        else:
            self._update_lr(iters - (self.warmup_iters if self.warmup is not None else 0), self.get_lr)

    def parameters(self):
        return {k: v for k, v in self.__dict__.items() if k != 'optimizer'}

    def load_parameters(self, data):
        if isinstance(data, dict):
            for k, v in data.items():
                if k in self.__dict__:
                    self.__dict__[k] = v


@SCHEDULERS.register_module()
class StepLR(WarmUpLR):
    def __init__(self, milestones, gamma=0.1, min_lr=None, **kwargs):
        if isinstance(milestones, list):
            assert all(s > 0 for s in milestones)
        elif isinstance(milestones, int):
            assert milestones > 0
        else:
            raise TypeError('"step" must be a list or integer')
        self.milestones, self.gamma, self.min_lr = milestones, gamma, min_lr
        super().__init__(**kwargs)

    def get_lr(self, base_lr, steps):
        if isinstance(self.milestones, int):
            exp = steps // self.milestones
        else:
            exp = len(self.milestones)
            for i, s in enumerate(self.milestones):
                if steps < s:
                    exp = i
                    break
        lr = base_lr * (self.gamma ** exp)
        return max(lr, self.min_lr) if self.min_lr is not None else lr


@SCHEDULERS.register_module()
class CosineAnnealingLR:
    """The SWA phase's schedule (lr_scheduler.py:274-320): built as ``(optimizer, min_lr | min_lr_ratio)`` and stepped
    with ``step(factor)``, ``factor = batch_idx / batches_per_epoch`` in [0, 1) -- one cosine half-wave from the base
    learning rate down to the target within EVERY epoch (runner.py:142-146), no warm-up."""

    def __init__(self, optimizer, min_lr=None, min_lr_ratio=None):
        self.optimizer, self.min_lr, self.min_lr_ratio = optimizer, min_lr, min_lr_ratio
        self.base_lr = optimizer.lr
        self.base_lr_pg = [pg.get("lr", optimizer.lr) for pg in optimizer.param_groups]
        self.step(0, 0)

    def get_lr(self, base_lr, factor):
        target = base_lr * self.min_lr_ratio if self.min_lr_ratio is not None else self.min_lr
        return target + 0.5 * (base_lr - target) * (math.cos(math.pi * factor) + 1)

    def step(self, factor, placeholder=None, **_):
        self.optimizer.lr = self.get_lr(self.base_lr, factor)
        for i, pg in enumerate(self.optimizer.param_groups):
            pg["lr"] = self.get_lr(self.base_lr_pg[i], factor)

    def parameters(self):
        return {k: v for k, v in self.__dict__.items() if k != 'optimizer'}

    def load_parameters(self, data):
        if isinstance(data, dict):
            for k, v in data.items():
                if k in self.__dict__:
                    self.__dict__[k] = v


@SCHEDULERS.register_module()
class ExpLR(WarmUpLR):
    def __init__(self, gamma, **kwargs):
        self.gamma = gamma
        super().__init__(**kwargs)

    def get_lr(self, base_lr, steps):
        return base_lr * self.gamma ** steps
