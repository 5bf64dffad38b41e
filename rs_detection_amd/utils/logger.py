"""Training-log hooks under the reference's names: ``RunLogger`` (what every config's ``logger=dict(type="RunLogger")``
builds), ``TextLogger`` and ``TensorboardLogger`` (/root/reference/python/jdet/utils/logger.py:10-68).

Same files and line format as the reference, so that tools that read a JDet ``work_dir`` keep working:

* ``<work_dir>/textlog/log_<YYYY_mm_dd_HH_MM_SS>.txt`` -- one line per logged step, ``<asctime>key:value,key:value,...``;
* ``<work_dir>/tensorboard/`` -- scalar events when a SummaryWriter is importable (``torch.utils.tensorboard`` or
  ``tensorboardX``); otherwise the same scalars as JSON lines in ``scalars.jsonl`` (this image ships neither writer);
* the console line of ``print_log`` (floats with 7 decimals, ``remain_time`` as ``[dD:hH:mM:sS]``).

Values may be 0-d tensors; they are read with ``.item()`` HERE, once per logged step (the Runner only calls ``log`` every
``log_interval`` steps and only on rank 0, after the cross-rank mean)."""
import json
import os
import time

from .registry import HOOKS, build_from_cfg

_NOT_SCALARS = ("iter", "epoch", "batch_idx", "times", "batch_size")


def _stamp():
    return time.asctime(time.localtime(time.time()))


def _plain(value):
    return value.item() if hasattr(value, "item") else value


@HOOKS.register_module()
class TextLogger:
    def __init__(self, work_dir):
        folder = os.path.join(os.path.abspath(work_dir), "textlog")
        os.makedirs(folder, exist_ok=True)
        self.path = os.path.join(folder, time.strftime("log_%Y_%m_%d_%H_%M_%S.txt", time.localtime()))
        self.log_file = open(self.path, "a")

    def log(self, data):
        self.log_file.write(_stamp() + ",".join("%s:%s" % kv for kv in data.items()) + "\n")
        self.log_file.flush()


@HOOKS.register_module()
class TensorboardLogger:
    def __init__(self, work_dir):
        self.dir = os.path.join(os.path.abspath(work_dir), "tensorboard")
        os.makedirs(self.dir, exist_ok=True)
        self.writer, self.lines = None, None
        for mod in ("torch.utils.tensorboard", "tensorboardX"):
            try:
                self.writer = __import__(mod, fromlist=["SummaryWriter"]).SummaryWriter(self.dir, flush_secs=10)
                break
            except Exception:
                continue
        if self.writer is None:
            self.lines = open(os.path.join(self.dir, "scalars.jsonl"), "a")

    def log(self, data):
        step = data["iter"]
        scalars = {k: v for k, v in data.items() if k not in _NOT_SCALARS and not isinstance(v, str)}
        if self.writer is not None:
            for k, v in scalars.items():
                self.writer.add_scalar(k, v, global_step=step)
        else:
            self.lines.write(json.dumps(dict(step=step, **scalars)) + "\n")
            self.lines.flush()


@HOOKS.register_module()
class RunLogger:
    def __init__(self, work_dir, loggers=("TextLogger", "TensorboardLogger")):
        self.loggers = [build_from_cfg(name, HOOKS, work_dir=work_dir) for name in loggers]

    @staticmethod
    def get_time(seconds):
        minutes, sec = divmod(int(seconds), 60)
        hours, minutes = divmod(minutes, 60)
        days, hours = divmod(hours, 24)
        return " [%dD:%dH:%dM:%dS] " % (days, hours, minutes, sec)

    def log(self, data, **kwargs):
        row = {k: _plain(v) for k, v in dict(data, **kwargs).items()}
        for sink in self.loggers:
            sink.log(row)
        self.print_log(row)

    def print_log(self, msg):
        print_record(msg)


def print_record(msg):
    """The console line of a record (a dict) or of a ready-made string."""
    if isinstance(msg, dict):
        parts = []
        for k, v in msg.items():
            if k == "remain_time":
                parts.append(" %s:%s" % (k, RunLogger.get_time(v)))
            elif isinstance(v, float):
                parts.append(" %s:%.7f" % (k, v))
            else:
                parts.append(" %s:%s" % (k, v))
        msg = ",".join(parts)
    print(_stamp(), msg)
