"""RunLogger / TextLogger / TensorboardLogger (reference: python/jdet/utils/logger.py:10-68) and the record the Runner
hands them (reference: python/jdet/runner/runner.py:151-171)."""
import glob
import json
import os
import re
import types

import torch

from rs_detection_amd.runner.runner import Runner
from rs_detection_amd.utils.registry import HOOKS, build_from_cfg


def test_run_logger_is_registered_and_writes_the_reference_files(tmp_path, capsys):
    for name in ("RunLogger", "TextLogger", "TensorboardLogger"):
        assert name in HOOKS
    lg = build_from_cfg(dict(type="RunLogger"), HOOKS, work_dir=str(tmp_path))          # what every config carries
    lg.log(dict(name="s2anet", lr=0.01, iter=50, epoch=0, batch_idx=49, batch_size=4, total_loss=torch.tensor(1.5),
                fps=80.0, eta="0:01:00"), loss_fam_cls=torch.tensor(0.25))
    files = glob.glob(str(tmp_path / "textlog" / "log_*.txt"))
    assert len(files) == 1 and re.search(r"log_\d{4}_\d\d_\d\d_\d\d_\d\d_\d\d\.txt$", files[0])
    line = open(files[0]).read().strip()
    # <asctime>key:value,key:value ... in insertion order, tensors read as Python numbers
    assert line.endswith("name:s2anet,lr:0.01,iter:50,epoch:0,batch_idx:49,batch_size:4,total_loss:1.5,fps:80.0,"
                         "eta:0:01:00,loss_fam_cls:0.25")
    tb = tmp_path / "tensorboard"
    assert tb.is_dir()
    if (tb / "scalars.jsonl").exists():        # no SummaryWriter in this image: the same scalars as JSON lines
        rec = json.loads((tb / "scalars.jsonl").read_text().strip())
        assert rec["step"] == 50 and rec["total_loss"] == 1.5 and "name" not in rec and "batch_size" not in rec
    out = capsys.readouterr().out
    assert " total_loss:1.5000000," in out and " iter:50," in out
    lg.print_log(dict(remain_time=93784))
    assert "remain_time: [1D:2H:3M:4S] " in capsys.readouterr().out
    only_text = build_from_cfg(dict(type="RunLogger", loggers=["TextLogger"]), HOOKS, work_dir=str(tmp_path / "b"))
    assert len(only_text.loggers) == 1


def test_runner_log_record_has_the_reference_keys(tmp_path):
    cfg = types.SimpleNamespace(name="demo", logger=dict(type="RunLogger", loggers=["TextLogger"]))
    opt = types.SimpleNamespace(cur_lr=lambda: 0.0025)
    r = types.SimpleNamespace(cfg=cfg, device=torch.device("cpu"), rank=0, world=1, max_epoch=12, max_iter=None, iter=100,
                              epoch=3, optimizer=opt, optimizer_swa=None, logger=None, work_dir=str(tmp_path))
    Runner._log_step(r, batch_idx=9, n_images=4, total=torch.tensor(2.0),
                     losses=dict(loss_odm_bbox=torch.tensor(0.5), loss_fam_cls=torch.tensor(1.5)), elapsed=2.0, n_batches=50)
    line = open(glob.glob(str(tmp_path / "textlog" / "log_*.txt"))[0]).read().strip()
    body = re.sub(r"^\w{3} \w{3} +\d+ \d\d:\d\d:\d\d \d{4}", "", line)          # strip the asctime stamp
    keys = [kv.split(":")[0] for kv in body.split(",")]
    assert keys[0] == "name"
    assert keys[1:9] == ["lr", "iter", "epoch", "batch_idx", "batch_size", "total_loss", "fps", "eta"]
    assert set(keys[9:]) == {"loss_fam_cls", "loss_odm_bbox"}
    assert "fps:20.0" in line and "total_loss:2.0" in line and "lr:0.0025" in line
    assert "eta:0:01:40" in line                     # (12 * 50 - 100) iterations left at 0.2 s each
    assert r.logger is not None                       # built once from cfg.logger, kept for the next record
