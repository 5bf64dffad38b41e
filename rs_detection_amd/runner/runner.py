"""Minimal Runner: build model / optimizer / scheduler from a JDet config and step it.

Counterpart of /root/reference/python/jdet/runner/runner.py:23-179 restricted to what the
hot path needs: ``train_step`` (forward, parse_losses, backward, grad all-reduce via DDP,
clip 35, SGD update, LR schedule -- :131-179) and ``test_time`` (:105-129: warm-up then timed
iterations on one cached batch, prints FPS), ``save`` / ``load`` / ``resume`` in the reference's checkpoint
layout (:251-290, runner/checkpoint.py), and the epoch loop ``run`` / ``train`` / ``val`` (:91-103, :131-208) over the
``dataset.train`` / ``dataset.val`` sections of the config (data/).  ``cfg.logger`` (RunLogger) writes the reference's text log; flip-test and the
tile-merge submission of ``test`` live in data/devkits.
"""
import os
import time

import torch

import rs_detection_amd.models  # noqa: F401  (registers everything)
from rs_detection_amd.optims import optimizer as _o, lr_scheduler as _s  # noqa: F401
from rs_detection_amd.utils import dist as rdist
from rs_detection_amd.utils import logger as _l  # (registers RunLogger / TextLogger / TensorboardLogger)
from rs_detection_amd.utils.general import parse_losses
from rs_detection_amd.utils.registry import MODELS, OPTIMS, SCHEDULERS, build_from_cfg


class Runner:
    def __init__(self, cfg, device=None, distributed=None, memory_format=None, amp_dtype=None, grad_dtype="auto",
                 bf16_params=None, fused_optimizer=True):
        self.cfg = cfg
        from rs_detection_amd.utils.miopen_db import use_packaged_miopen_db
        use_packaged_miopen_db()  # tuned MIOpen solver records of the shipped configs (before the first convolution)
        self.rank, self.local_rank, self.world = rdist.env_world()
        if device is None:
            device = torch.device("cuda", self.local_rank) if torch.cuda.is_available() else torch.device("cpu")
        self.device = device
        self.model = build_from_cfg(cfg.model, MODELS).to(device)
        if memory_format is not None and memory_format != "trunk_channels_last":
            # only rank-4 parameters have a channels_last form (the ARF weight is rank 5)
            for p in self.model.parameters():
                if p.dim() == 4:
                    p.data = p.data.contiguous(memory_format=memory_format)
        if memory_format == "trunk_channels_last":       # only the backbone runs NHWC (ResNet.set_channels_last)
            memory_format = None
            if hasattr(self.model, "backbone") and hasattr(self.model.backbone, "set_channels_last"):
                self.model.backbone.set_channels_last(True)
        self.memory_format = memory_format
        self.amp_dtype = amp_dtype
        # bf16 PARAMETERS (launch diet of the bf16 step): the weights of the plain convolutions and linear
        # layers live in bf16 in the model, their fp32 masters and momenta in FusedSGD (csrc/optim.hip: clip +
        # SGD + both copies in two launches).  Removes, per step, one fp32->bf16 cast per weight (autocast), one
        # bf16->fp32 cast per weight gradient and the foreach passes of clip_grad_norm_ / SGD (~250 launches of ~1 700).
        # BatchNorm parameters, the ARF weight of ORConv2d (fp32 kernels) and every other parameter stay fp32.
        # On by default wherever it applies (bf16 autocast + SGD, no SWA phase): what bench.py times.  NOTE for exporters:
        # ``model.state_dict()`` of such a Runner holds ROUNDED bf16 weights -- ``Runner.state_dict_fp32()`` / ``save()``
        # return the unrounded fp32 values (README.md); ``bf16_params=False`` keeps fp32 weights.
        if bf16_params is None:
            bf16_params = True
        # Not with an SWA phase: ``optimizer_swa`` is a plain optimizer over the same parameters and would update the bf16
        # copies while FusedSGD's fp32 masters went stale (and the checkpoint, which stores the masters, lost the phase).
        self.bf16_params = bool(bf16_params) and amp_dtype == torch.bfloat16 and device.type == "cuda" and \
            bool(cfg.optimizer) and cfg.optimizer.get("type") == "SGD" and not getattr(cfg, "optimizer_swa", None)
        frozen_masters = {}
        if self.bf16_params:
            for mname, m in self.model.named_modules():
                if type(m) in (torch.nn.Conv2d, torch.nn.Linear):
                    # weights only (frozen stages too: autocast casts those every step as well); the biases stay fp32
                    # -- the fused bias + ReLU tails (ops/bn_act.bias_act) take fp32 per-channel parameters.  A frozen
                    # weight has no optimizer state to hold its fp32 original, so the Runner keeps it for checkpoints.
                    if not m.weight.requires_grad:
                        frozen_masters[(mname + "." if mname else "") + "weight"] = m.weight.detach().float().clone()
                    m.weight.data = m.weight.data.to(torch.bfloat16)
        params = [p for p in self.model.parameters() if p.requires_grad]
        opt_cfg = cfg.optimizer
        # The two-launch clip + SGD step (optims.FusedSGD, csrc/optim.hip) serves fp32 parameters as well: same
        # arithmetic as torch's clip_grad_norm_ + SGD (tests/test_gpu_optim.py), ~15 foreach launches fewer per step.
        # ``fused_optimizer=False`` keeps torch.optim.SGD / AdamW (what the fused steps are tested against).
        fused_ok = (device.type == "cuda" and bool(cfg.optimizer) and cfg.optimizer.get("type") == "SGD"
                    and not cfg.optimizer.get("dampening", 0) and not cfg.optimizer.get("nesterov", False)
                    and bool(fused_optimizer))
        if self.bf16_params or fused_ok:
            opt_cfg = dict(cfg.optimizer, type="FusedSGD")
        # AdamW (configs/orcnn): the same two launches with the AdamW update (optims.FusedAdamW) instead
        # of torch.optim.AdamW (foreach: ~130 launches and 2.9 ms per step on VAN-B3)
        elif (device.type == "cuda" and bool(cfg.optimizer) and cfg.optimizer.get("type") == "AdamW"
              and bool(fused_optimizer)):
            opt_cfg = dict(cfg.optimizer, type="FusedAdamW")
        self.optimizer = build_from_cfg(opt_cfg, OPTIMS, params=params) if cfg.optimizer else None
        if frozen_masters and self.optimizer is not None:
            self.optimizer.frozen_masters = frozen_masters
        self.scheduler = build_from_cfg(cfg.scheduler, SCHEDULERS, optimizer=self.optimizer) \
            if (cfg.scheduler and self.optimizer) else None
        # the SWA phase (runner.py:51-53): its own optimizer + per-epoch cosine schedule over the same parameters
        self.swa_start_epoch = getattr(cfg, "swa_start_epoch", None)
        self.optimizer_swa = build_from_cfg(cfg.optimizer_swa, OPTIMS, params=params) \
            if getattr(cfg, "optimizer_swa", None) else None
        self.scheduler_swa = build_from_cfg(cfg.scheduler_swa, SCHEDULERS, optimizer=self.optimizer_swa) \
            if (getattr(cfg, "scheduler_swa", None) and self.optimizer_swa) else None
        if distributed is None:
            distributed = self.world > 1
        # gradient buckets travel in bf16 when the step computes in bf16 (BASELINE configs[2..4]); fp32 otherwise
        self.grad_dtype = (amp_dtype if amp_dtype == torch.bfloat16 else None) if grad_dtype == "auto" else grad_dtype
        # Data parallelism: the bucketed gradient mean of utils/reducer.py (one multi-tensor copy + one all-reduce per
        # 64 MB bucket, flushed from inside backward; its module docstring says why not torch's DDP on a host-paced step).
        #   distributed=True / "force"   own reducer ("force": also in a one-rank group -- RCCL on one GPU: tests, bench)
        #   distributed="ddp" / "ddp-force"   torch.nn.parallel.DistributedDataParallel, what the reducer is tested against
        self.reducer = None
        self.ddp = self.model
        if distributed in ("ddp", "ddp-force"):
            self.ddp = rdist.wrap_ddp(self.model, device, grad_dtype=self.grad_dtype, force=(distributed == "ddp-force"))
        elif distributed and torch.distributed.is_available() and torch.distributed.is_initialized() and \
                (self.world > 1 or distributed == "force"):
            from rs_detection_amd.utils.reducer import GradReducer
            self.reducer = GradReducer(self.model)       # (broadcasts rank 0's parameters and buffers)
            if frozen_masters and self.world > 1:
                # the fp32 originals of the frozen bf16 weights were taken BEFORE that broadcast: rank 0's as well
                for k in sorted(frozen_masters):
                    torch.distributed.broadcast(frozen_masters[k], 0)
        self.iter, self.epoch = 0, 0
        self.max_epoch = cfg.max_epoch if hasattr(cfg, "max_epoch") else None
        self.max_iter = cfg.max_iter if hasattr(cfg, "max_iter") else None
        self.train_dataset = self.val_dataset = self.test_dataset = None
        self.work_dir = None
        self.logger = None          # built from cfg.logger on the first record, once work_dir is known (rank 0)

    def train_step(self, images, targets, swa_factor=None):
        """One iteration (:138-150).  ``swa_factor`` = batch_idx / batches_per_epoch switches to the SWA optimizer and
        its schedule (:142-146); None = the main optimizer + ``scheduler.step(iter, epoch)``."""
        # model.train() walks ~570 modules and re-applies norm_eval / frozen stages: 2.3 ms of host time, so it runs on
        # the first step and whenever something (val(), a caller) has put the model into eval mode, not on every step
        if not (self.model.training and getattr(self, "_train_mode_applied", False)):
            self.model.train()
            self._train_mode_applied = True
        if self.memory_format is not None:
            images = images.contiguous(memory_format=self.memory_format)
        if self.amp_dtype is not None:
            with torch.autocast(device_type=self.device.type, dtype=self.amp_dtype):
                losses = self.ddp(images, targets)
        else:
            losses = self.ddp(images, targets)
        total, parsed = parse_losses(losses)
        swa = swa_factor is not None and self.optimizer_swa is not None
        opt = self.optimizer_swa if swa else self.optimizer
        opt.zero_grad(set_to_none=True)
        if self.reducer is not None:
            self.reducer.begin_step()    # every gradient None: the buckets may leave from inside backward (utils/reducer.py)
        total.backward()
        if self.reducer is not None:
            self.reducer.reduce()        # gradient mean over the ranks (most of it already in flight: utils/reducer.py)
        opt.step()
        if swa:
            if self.scheduler_swa is not None:
                self.scheduler_swa.step(swa_factor)
        elif self.scheduler is not None:
            self.scheduler.step(self.iter, self.epoch, by_epoch=True)
        self.iter += 1
        return total, parsed

    # ---- epoch loop (:86-103, :131-208) -----------------------------------------------------
    def build_datasets(self, work_dir=None):
        """``dataset.train`` / ``dataset.val`` of the config -> datasets, sharded over the ranks."""
        import rs_detection_amd.data  # noqa: F401  (registers DATASETS / TRANSFORMS)
        from rs_detection_amd.utils.registry import DATASETS
        ds = self.cfg.dataset if hasattr(self.cfg, "dataset") and self.cfg.dataset else {}
        if ds.get("train"):
            self.train_dataset = build_from_cfg(ds["train"], DATASETS)
            self.train_dataset.set_shard(self.rank, self.world)
        if ds.get("val"):
            self.val_dataset = build_from_cfg(ds["val"], DATASETS)
        if ds.get("test"):
            self.test_dataset = build_from_cfg(ds["test"], DATASETS)
        self.work_dir = work_dir
        return self

    @property
    def finish(self):
        if self.max_epoch:
            return self.epoch >= self.max_epoch
        return self.max_iter is not None and self.iter >= self.max_iter

    def train(self, log_interval=50):
        """One epoch over ``train_dataset`` (:131-179)."""
        from rs_detection_amd.data.loader import prefetch_to_device
        self.train_dataset.set_epoch(self.epoch)
        self.model.train()                                        # once per epoch, as :133
        self._train_mode_applied = True
        start, last = time.time(), None
        swa = self.swa_start_epoch is not None and self.epoch >= self.swa_start_epoch and self.optimizer_swa is not None
        n_batches = max(len(self.train_dataset._indices()) // max(self.train_dataset.batch_size, 1), 1)
        # decode / transforms in the dataset's worker processes, collate + pinned H2D copy on a side stream, two
        # batches ahead of the step (data/loader.py)
        for batch_idx, (images, targets) in enumerate(prefetch_to_device(self.train_dataset, self.device)):
            # train_step advances self.iter
            total, losses = self.train_step(images, targets, swa_factor=batch_idx / n_batches if swa else None)
            last = total
            if log_interval and self.iter % log_interval == 0:
                self._log_step(batch_idx, len(targets), total, losses, time.time() - start, n_batches, swa)
            if self.finish:
                break
        self.epoch += 1
        return last

    def _log_step(self, batch_idx, n_images, total, losses, elapsed, n_batches, swa=False):
        """The reference's log record (runner.py:151-171): name, lr, iter, epoch, batch_idx, batch_size, total_loss, fps,
        eta and the loss terms, averaged over the ranks (``sync``), handed on rank 0 to the logger the config names
        (``logger=dict(type="RunLogger")`` -> ``<work_dir>/textlog``, ``<work_dir>/tensorboard``, a console line).
        Without a ``work_dir`` there are no files: the console line alone."""
        import datetime
        scal = dict(losses)
        scal["total_loss"] = total
        scal = rdist.sync_mean(scal, self.device)          # every rank takes part in the mean; one host read per record
        if self.rank != 0:
            return
        batch_size = n_images * self.world
        if self.max_epoch:
            total_iter = self.max_epoch * n_batches
        else:
            total_iter = self.max_iter or self.iter
        eta = max(total_iter - self.iter, 0) * elapsed / (batch_idx + 1)
        data = dict(name=getattr(self.cfg, "name", None), lr=(self.optimizer_swa if swa else self.optimizer).cur_lr(),
                    iter=self.iter, epoch=self.epoch, batch_idx=batch_idx, batch_size=batch_size,
                    total_loss=scal.pop("total_loss"), fps=batch_size * (batch_idx + 1) / max(elapsed, 1e-9),
                    eta=str(datetime.timedelta(seconds=int(eta))))
        data.update(scal)
        if self.logger is None and self.work_dir and getattr(self.cfg, "logger", None):
            from rs_detection_amd.utils.registry import HOOKS
            self.logger = build_from_cfg(self.cfg.logger, HOOKS, work_dir=self.work_dir)
        if self.logger is not None:
            self.logger.log(data)
        else:
            _l.print_record(data)

    @torch.no_grad()
    def _predict_dataset(self, dataset, flip_test=()):
        """Predictions of this rank's shard of ``dataset`` (every image exactly once over the ranks), gathered on every
        rank: [((polys, scores, labels) as NumPy, target)] in dataset order of each shard.  ``flip_test`` adds the
        predictions of the 'H' / 'V' / 'HV' flipped images, tagged ``flip_mode`` (:221-235)."""
        import copy
        from rs_detection_amd.data import batch_to_device
        dataset.set_shard(self.rank, self.world, keep_all=True)
        results = []

        def to_np(pred):
            polys, scores, labels = pred
            return polys.double().cpu().numpy(), scores.cpu().numpy(), labels.long().cpu().numpy()

        for images, targets in dataset:
            timg, ttg = batch_to_device(images, targets, self.device)
            for pred, t in zip(self.predict(timg, ttg), targets):
                results.append((to_np(pred), t))
            for mode in flip_test:
                dims = {"H": (3,), "V": (2,), "HV": (2, 3)}[mode]
                for pred, t in zip(self.predict(torch.flip(timg, dims), ttg), targets):
                    t = copy.deepcopy(t)
                    t["flip_mode"] = mode
                    results.append((to_np(pred), t))
        gathered = rdist.gather_objects(results)    # also the barrier that keeps the ranks together after a long eval
        return [r for part in gathered for r in part]

    def val(self):
        """:196-208: predictions of ``val_dataset`` -> ``val_dataset.evaluate`` (DOTA mAP on the GPU).  The reference
        evaluates on rank 0 alone while the others wait; here the images are sharded over the ranks and gathered, so
        no rank sits in the next epoch's all-reduce while another is still evaluating."""
        if self.val_dataset is None:
            return None
        results = self._predict_dataset(self.val_dataset)
        if self.rank != 0:
            return None
        return self.val_dataset.evaluate(results, self.work_dir, self.epoch, device=self.device)

    def test(self, name=None):
        """:210-249: predictions of ``test_dataset`` (plus the 'H' / 'V' / 'HV' flipped passes named by
        ``cfg.flip_test``), pickled to ``<work_dir>/test/test_<epoch>.pkl`` like the reference, and -- for an
        ImageDataset of tiles -- merged into whole-image detections and the submission archive
        (data/devkits/data_merge.py; ``cfg.merge_nms_threshold_type`` selects the per-class NMS thresholds)."""
        import os
        import pickle
        if self.test_dataset is None:
            return None
        flips = list(getattr(self.cfg, "flip_test", None) or [])
        assert all(m in ("H", "V", "HV") for m in flips), flips
        results = self._predict_dataset(self.test_dataset, flips)
        if self.rank != 0:
            return None
        out = dict(results=results)
        if self.work_dir:
            os.makedirs(os.path.join(self.work_dir, "test"), exist_ok=True)
            pkl = os.path.join(self.work_dir, "test", "test_%d.pkl" % self.epoch)
            with open(pkl, "wb") as f:
                pickle.dump(results, f)
            out["pkl"] = pkl
            ds_cfg = (self.cfg.dataset or {}).get("test", {}) if hasattr(self.cfg, "dataset") else {}
            if ds_cfg.get("type") == "ImageDataset":
                from rs_detection_amd.data.devkits.data_merge import data_merge_result
                nm = name or ("%s_epoch%d" % (getattr(self.cfg, "name", None) or "run", self.epoch))
                out["submission"] = data_merge_result(
                    results, self.work_dir, self.epoch, nm, self.test_dataset.dataset_type,
                    ds_cfg.get("images_dir", ""), device=self.device,
                    nms_threshold_type=getattr(self.cfg, "merge_nms_threshold_type", None) or 0)
        return out

    def run(self, checkpoint_interval=None, eval_interval=None, log_interval=50):
        """:91-103: train epochs until ``max_epoch``; checkpoint / evaluate at the config's intervals."""
        import os
        ci = checkpoint_interval if checkpoint_interval is not None else getattr(self.cfg, "checkpoint_interval", None)
        ei = eval_interval if eval_interval is not None else getattr(self.cfg, "eval_interval", None)
        evals = {}
        while not self.finish:
            self.train(log_interval)
            if ci and self.epoch % ci == 0 and self.work_dir:
                os.makedirs(os.path.join(self.work_dir, "checkpoints"), exist_ok=True)
                self.save(os.path.join(self.work_dir, "checkpoints", "ckpt_%d.pkl" % self.epoch))
            if ei and self.epoch % ei == 0:
                evals[self.epoch] = self.val()
        return evals

    # ---- checkpoints (:251-290) -----------------------------------------------------------
    def save(self, path):
        from .checkpoint import save_checkpoint
        if self.rank != 0:
            return None
        return save_checkpoint(path, self.model, self.optimizer, self.scheduler,
                               meta=dict(epoch=self.epoch, iter=self.iter, config=dict(self.cfg) if hasattr(self.cfg, "keys") else None))

    def state_dict_fp32(self):
        """``model.state_dict()`` with every floating-point entry in fp32 and UNROUNDED: where the Runner holds a weight
        in bf16 (``bf16_params``) the value comes from the optimizer's fp32 master (trainable weights) or the Runner's
        kept original (frozen weights).  What an exporter / external evaluator should read instead of
        ``model.state_dict()``; ``save()`` writes the same values."""
        sd = {k: (v.detach().float() if v.is_floating_point() else v.detach()) for k, v in self.model.state_dict().items()}
        if self.optimizer is not None and hasattr(self.optimizer, "master_state_dict"):
            for k, v in self.optimizer.master_state_dict(self.model).items():
                sd[k] = v.detach().float()
        return sd

    def load(self, load_path, model_only=False):
        from .checkpoint import read_checkpoint, model_parameters, load_parameters
        data = read_checkpoint(load_path)
        if not model_only and isinstance(data, dict):
            meta = data.get("meta", dict())
            self.epoch, self.iter = meta.get("epoch", self.epoch), meta.get("iter", self.iter)
            if self.scheduler is not None:
                self.scheduler.load_parameters(data.get("scheduler", dict()))
            opt = data.get("optimizer")
            if self.optimizer is not None and isinstance(opt, dict) and "param_groups" in opt:
                import numpy as np
                def back(o):
                    if isinstance(o, np.ndarray):
                        return torch.from_numpy(o)
                    if isinstance(o, dict):
                        return {k: back(v) for k, v in o.items()}
                    if isinstance(o, list):
                        return [back(v) for v in o]
                    return o
                self.optimizer.load_state_dict(back(opt))
        out = load_parameters(self.model, model_parameters(data))
        if hasattr(self.optimizer, "set_masters"):        # bf16 model weights: the file's fp32 values are the masters
            self.optimizer.set_masters(self.model, model_parameters(data))
        return out

    resume = load

    @torch.no_grad()
    def predict(self, images, targets):
        """Inference in the dtype the model trains in: fp32, or bf16 autocast (with bf16 parameters the convolutions
        need it: their weights ARE bf16)."""
        if self.model.training:         # (only on the first call of an evaluation: walking 8 000 modules costs 1.7 ms, 15 % of
            self.model.eval()           #  a B = 1 call)
        if self.memory_format is not None:
            images = images.contiguous(memory_format=self.memory_format)
        if self.amp_dtype is not None:
            with torch.autocast(device_type=self.device.type, dtype=self.amp_dtype):
                return self.model(images, targets)
        return self.model(images, targets)

    def test_time(self, images, targets, warmup=10, iters=100):
        for _ in range(warmup):
            self.train_step(images, targets)
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(iters):
            self.train_step(images, targets)
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        dt = time.time() - t0
        fps = images.shape[0] * self.world * iters / dt
        print("FPS:", fps)
        return fps
