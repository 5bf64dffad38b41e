"""Generates tests/golden/host_logic.npz by running the reference's own pure-Python host logic (build container only):
  jdet/optims/lr_scheduler.py  StepLR (linear warm-up) and CosineAnnealingLR  :8-60,196-236,274-320
  jdet/data/devkits/voc_eval.py  voc_ap (07 metric and area)                  :39-71
  jdet/data/devkits/result_merge.py  py_cpu_nms, poly2origpoly               :140-194
  jdet/data/devkits/data_merge.py  flip_box                                  :14-27
Imported from where they lie with the same inert placeholders for jittor / cv2 / shapely as make_transforms_golden.py
(none of these functions touches them)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_transforms_golden import load_reference  # noqa: E402


class _Opt:
    """The two attributes the reference schedulers touch on a Jittor optimizer."""
    def __init__(self, lr, groups):
        self.lr, self.param_groups = lr, [dict(g) for g in groups]


def main():
    import importlib
    import types
    load_reference()
    for name, sub in (("jdet.optims", "/optims"), ("jdet.data.devkits", "/data/devkits")):
        mod = types.ModuleType(name)
        mod.__path__ = ["/root/reference/python/jdet" + sub]
        sys.modules[name] = mod
    for m in ("tqdm", "zipfile36"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules["tqdm"].tqdm = lambda x, **k: x
    cfgmod = importlib.import_module("jdet.config.config")     # the real loader: `from jdet.config import get_cfg`
    sys.modules["jdet.config"].get_cfg = cfgmod.get_cfg
    S = importlib.import_module("jdet.optims.lr_scheduler")
    out = {}
    # ---- StepLR with linear warm-up, by_epoch=True: the S2ANet schedule
    opt = _Opt(0.0025, [dict(lr=0.0025), dict(lr=0.005)])
    sch = S.StepLR(optimizer=opt, milestones=[7, 10], gamma=0.1, warmup='linear', warmup_iters=500, warmup_ratio=1.0 / 3)
    grid = [(it, ep) for ep in range(12) for it in (ep * 600 + k for k in (0, 1, 250, 499, 500, 599))]
    vals = []
    for it, ep in grid:
        sch.step(it, ep, by_epoch=True)
        vals.append([it, ep, opt.lr, opt.param_groups[0]["lr"], opt.param_groups[1]["lr"]])
    out["steplr"] = np.array(vals, np.float64)
    # ---- CosineAnnealingLR (the SWA phase)
    opt = _Opt(1e-4, [dict(lr=1e-4)])
    cs = S.CosineAnnealingLR(opt, min_lr=1e-6)
    fac = np.linspace(0, 0.999, 40)
    vals = []
    for f in fac:
        cs.step(float(f))
        vals.append([f, opt.lr, opt.param_groups[0]["lr"]])
    out["cosine"] = np.array(vals, np.float64)
    # ---- voc_ap
    V = importlib.import_module("jdet.data.devkits.voc_eval")
    rng = np.random.default_rng(0)
    for i in range(4):
        n = int(rng.integers(5, 200))
        tp = (rng.random(n) < 0.6).astype(np.float64)
        fp = 1 - tp
        rec = np.cumsum(tp) / max(tp.sum() + rng.integers(0, 5), 1)
        prec = np.cumsum(tp) / np.maximum(np.cumsum(tp) + np.cumsum(fp), np.finfo(np.float64).eps)
        out["ap%d_rec" % i], out["ap%d_prec" % i] = rec, prec
        out["ap%d" % i] = np.array([V.voc_ap(rec, prec, True), V.voc_ap(rec, prec, False)])
    # ---- result_merge helpers
    R = importlib.import_module("jdet.data.devkits.result_merge")
    c = rng.uniform(0, 300, (80, 2))
    wh = rng.uniform(10, 90, (80, 2))
    dets = np.concatenate([c - wh / 2, c + wh / 2, rng.uniform(0, 1, (80, 1))], 1)
    out["hbb_dets"] = dets
    for thr in (0.1, 0.3, 0.5):
        out["hbb_keep_%02d" % int(thr * 10)] = np.array(R.py_cpu_nms(dets.copy(), thr), np.int64)
    poly = rng.uniform(0, 1024, (5, 8))
    out["o2p_in"] = poly
    out["o2p_out"] = np.array([R.poly2origpoly(list(p), 824, 1648, "0.5") for p in poly])
    D = importlib.import_module("jdet.data.devkits.data_merge")
    box = rng.uniform(0, 100, 8)
    out["flip_in"] = box
    for mode in ("H", "V", "HV"):
        out["flip_" + mode] = np.array(D.flip_box(list(box), dict(flip_mode=mode, ori_img_size=(120, 90))))
    out["provenance"] = np.array("reference lr_scheduler.py / voc_eval.py / result_merge.py / data_merge.py run in the "
                                 "build container through tests/golden/make_host_golden.py")
    np.savez_compressed(os.path.join(HERE, "host_logic.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
