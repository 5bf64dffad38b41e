"""Learning-rate schedules of the reference's config surface (``scheduler = dict(type='StepLR', warmup='linear',
warmup_iters=500, warmup_ratio=1/3, milestones=[7, 10])``, ``scheduler_swa = dict(type='CosineAnnealingLR', ...)``).

Each schedule is a PURE function ``multiplier-or-value = f(base, progress)``; one small adapter (``_Schedule``) owns the
optimizer handle, picks the clock (iteration during warm-up, epoch or post-warm-up iteration after it) and writes the
result into every parameter group.  What the functions compute is the reference's
/root/reference/python/jdet/optims/lr_scheduler.py:8-60 (warm-up), :196-236 (step decay), :274-320 (cosine); their
values are pinned by arrays the reference itself produced (tests/golden/host_logic.npz, tests/test_host_golden_cpu.py).
The Runner calls ``step(iter, epoch, by_epoch=True)`` once per iteration, the SWA phase ``step(batch_idx / batches)``."""
import bisect
import math

from rs_detection_amd.utils.registry import SCHEDULERS


# ---------------------------------------------------------------- pure schedule functions

def warmup_scale(kind, ratio, it, n_iters):
    """Multiplier on the base rate while ``it < n_iters``: 'constant' holds ``ratio``, 'linear' walks ratio -> 1,
    'exp' walks ratio -> 1 geometrically."""
    left = 1.0 - it / n_iters                   # share of the warm-up still ahead
    table = {"constant": lambda: ratio,
             "linear": lambda: 1 - left * (1 - ratio),
             "exp": lambda: ratio ** left}
    if kind not in table:
        raise ValueError(kind)
    return table[kind]()


def step_decay(base, t, milestones, gamma, floor=None):
    """``base * gamma ** (number of milestones already passed at clock t)``; an int milestone = every that many ticks."""
    passed = t // milestones if isinstance(milestones, int) else bisect.bisect_right(list(milestones), t)
    lr = base * (gamma ** passed)
    return lr if floor is None else max(lr, floor)


def exp_decay(base, t, gamma):
    return base * gamma ** t


def cosine_to(base, target, frac):
    """Half cosine wave from ``base`` (frac = 0) to ``target`` (frac = 1)."""
    return target + 0.5 * (base - target) * (math.cos(math.pi * frac) + 1)


# ---------------------------------------------------------------- the stateful adapter

class _Schedule:
    """Holds the optimizer and its base rates; subclasses provide ``value(base, t)``.  ``parameters()`` /
    ``load_parameters()`` are the checkpoint surface (everything but the optimizer handle)."""

    def _bind(self, optimizer):
        self.optimizer = optimizer
        self.base_lr = optimizer.lr
        self.base_lr_pg = [pg.get("lr", optimizer.lr) for pg in optimizer.param_groups]

    def _write(self, f):
        self.optimizer.lr = f(self.base_lr)
        for pg, base in zip(self.optimizer.param_groups, self.base_lr_pg):
            pg["lr"] = f(base)

    def parameters(self):
        return {k: v for k, v in vars(self).items() if k != "optimizer"}

    def load_parameters(self, data):
        if isinstance(data, dict):
            vars(self).update({k: v for k, v in data.items() if k in vars(self)})


@SCHEDULERS.register_module()
class WarmUpLR(_Schedule):
    """Warm-up over the first ``warmup_iters`` iterations, then ``value(base, clock)`` -- constant here."""

    def __init__(self, optimizer, warmup_ratio=1.0 / 3, warmup_iters=500, warmup=None):
        self.warmup_ratio, self.warmup_iters, self.warmup = warmup_ratio, warmup_iters, warmup
        self._bind(optimizer)
        self.step(0, 0)

    def value(self, base, t):
        return base

    # names the reference's subclasses override / call; kept as thin views of the functions above
    def get_lr(self, base, t):
        return self.value(base, t)

    def get_warmup_lr(self, base, it):
        return base * warmup_scale(self.warmup, self.warmup_ratio, it, self.warmup_iters)

    def step(self, iters, epochs, by_epoch=True):
        warm = self.warmup is not None
        if warm and iters < self.warmup_iters:
            return self._write(lambda b: self.get_warmup_lr(b, iters))
        t = epochs if by_epoch else iters - (self.warmup_iters if warm else 0)
        self._write(lambda b: self.value(b, t))


@SCHEDULERS.register_module()
class StepLR(WarmUpLR):
    def __init__(self, milestones, gamma=0.1, min_lr=None, **kwargs):
        ok = milestones > 0 if isinstance(milestones, int) else \
            isinstance(milestones, list) and all(s > 0 for s in milestones)
        if not isinstance(milestones, (int, list)):
            raise TypeError('"step" must be a list or integer')
        assert ok, milestones
        self.milestones, self.gamma, self.min_lr = milestones, gamma, min_lr
        super().__init__(**kwargs)

    def value(self, base, t):
        return step_decay(base, t, self.milestones, self.gamma, self.min_lr)


@SCHEDULERS.register_module()
class ExpLR(WarmUpLR):
    def __init__(self, gamma, **kwargs):
        self.gamma = gamma
        super().__init__(**kwargs)

    def value(self, base, t):
        return exp_decay(base, t, self.gamma)


@SCHEDULERS.register_module()
class CosineAnnealingLR(_Schedule):
    """The SWA phase's schedule: built as ``(optimizer, min_lr | min_lr_ratio)`` and stepped with ``step(factor)``,
    ``factor = batch_idx / batches_per_epoch`` in [0, 1) -- one cosine half-wave from the base rate down to the target
    within EVERY epoch (reference runner.py:142-146), no warm-up."""

    def __init__(self, optimizer, min_lr=None, min_lr_ratio=None):
        self.min_lr, self.min_lr_ratio = min_lr, min_lr_ratio
        self._bind(optimizer)
        self.step(0, 0)

    def get_lr(self, base, factor):
        target = self.min_lr if self.min_lr_ratio is None else base * self.min_lr_ratio
        return cosine_to(base, target, factor)

    def step(self, factor, placeholder=None, **_):
        self._write(lambda b: self.get_lr(b, factor))
