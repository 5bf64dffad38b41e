"""GPU: regression guards for the two numbers that decide multi-GPU scaling with one Python process per GPU --
kernel launches per step and host<->device synchronisations per step ("Missing" #7).  Each case is a child
process with the launch-counter shim preloaded (tests/tools/launch_counter.cpp: counts hipLaunchKernel /
hip(Ext)ModuleLaunchKernel / ... whoever calls them) running the BASELINE config at the bench's own shapes.

Budgets = the measured counts of the current code + ~3 % (profiles/r05_*: see DESIGN §R5); a change that adds launches
or an `.item()` to a step fails here, not in a profile three rounds later.  Lower them when a step gets leaner."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.launcher]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (model, dtype) -> (max launches + fills per step, max syncs per step)
# measured (round 5, call 3): s2anet f32 844 + 13 fills, bf16 722 + 17, orcnn 4313 + 18 with 8 syncs before the fixed-size
# sampling path; the Oriented R-CNN step has had no synchronisation since
# round 6: the VAN Block as one node of 42 launches (csrc/van_block.hip): orcnn 4313 -> 3560 + 12 fills; the heads'
# control path as kernels (csrc/orpn.hip: proposals, samplers, RPN losses on the samples, RoI targets): -> 2165 + 12; the
# block's row folds and depthwise finishing passes as one launch each: -> 2013 + 12
# end of round 6, measured: s2anet f32 834 + 13 fills, bf16 600 + 17, orcnn 1975 + 14; then the AlignConv backwards of the
# five levels share one col2im index build (25 launches + 5 fills -> 9 + 1): s2anet f32 and bf16 -24; one weight cast per
# pass, bf16 gradient stored by the gather, channels_last weight gradient (bf16 -14); refine + offset of all levels in one
# launch (bf16 -9, f32 -3)
# measured on the final code of round 6: s2anet f32 813 + 9 fills, bf16 558 + 13, orcnn 1916 + 9 (1961 + 12 before the
# one-index RoI extractor backward and the fused RPN bias + ReLU)
BUDGET = {("s2anet", "f32"): (840, 0), ("s2anet", "bf16"): (590, 0), ("orcnn", "f32"): (1965, 0)}


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("lc") / "liblaunch_counter.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "tools", "launch_counter.cpp"), "-o", out, "-ldl"])
    return out


@pytest.mark.timeout(900)
@pytest.mark.parametrize("model,dtype", sorted(BUDGET))
def test_step_launch_and_sync_budget(model, dtype, shim):
    if torch.cuda.device_count() == 0:
        pytest.skip("no GPU")
    env = dict(os.environ, LD_PRELOAD=shim, RSDET_LAUNCH_COUNTER=shim)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "step_guard_worker.py"), model, dtype, "1024"],
                       env=env, capture_output=True, text=True, timeout=850)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    print(r)
    max_launches, max_syncs = BUDGET[(model, dtype)]
    assert r["sync_error"] is None, "a synchronising call inside the %s %s step: %s" % (model, dtype, r["sync_error"])
    assert r["syncs"] <= max_syncs, r
    assert 100 < r["launches"] + r["fills"] <= max_launches, r
