"""GPU: does the assembled detector LEARN?  The stand-in for the reference's accuracy gate (mAP at 12 epochs of DOTA,
which is not available offline): train the real model classes from random initialisation on a rendered DOTA-format set
(data/synthetic.py: every gt painted as a filled rotated rectangle in its class colour) at the reference's optimiser
settings and require (i) the loss to fall window over window and (ii) DOTA mAP (polygon IoU 0.5, VOC07-style AP, the
same evaluate() as a real run) on the training images well above chance.  Measured on MI355X (profiles/scripts/
learn_proof.py): S2ANet-R50-FPN fp32, lr 0.0025: loss 3.7 -> 0.20 and mAP 0.94 after 600 iterations (23 s)."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lp():
    spec = importlib.util.spec_from_file_location("learn_proof", os.path.join(ROOT, "profiles", "scripts", "learn_proof.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _windows(losses, n=4):
    w = len(losses) // n
    return [float(np.median(losses[i * w:(i + 1) * w])) for i in range(n)]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("model,dtype,lr,iters,min_map", [("s2anet", "f32", 0.0025, 500, 0.5),
                                                          ("s2anet", "bf16", 0.0025, 500, 0.5),
                                                          ("orcnn", "f32", 0.0004, 600, 0.3)])
def test_model_learns_the_rendered_set(cuda, model, dtype, lr, iters, min_map):
    lp = _lp()
    r = lp.make_runner(model, dtype, lr=lr)
    losses = lp.train(r, iters, log=0)
    assert np.isfinite(losses).all()
    w = _windows(losses)
    assert w[1] < w[0] and w[2] < w[1] and w[3] < w[2] * 1.05, w      # falls window over window
    assert w[3] < 0.35 * w[0], w
    ev = r.val()
    print(model, dtype, "loss windows", [round(x, 3) for x in w], "mAP %.3f" % ev["eval/0_meanAP"])
    assert ev["eval/0_meanAP"] > min_map, ev
