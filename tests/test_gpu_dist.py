"""GPU: the N>1 path on real hardware with the REAL models.  The box has one GPU, so two ranks share cuda:0 and
talk through gloo (RCCL refuses two ranks on one device; with >= 2 GPUs the same test runs over RCCL).  The ranks are
child processes started by rs_detection_amd.utils.dist.launch_ranks -- nothing is exec'ed over or forked from a
process that has initialised HIP -- and conftest.py schedules this module before every other GPU test so that the
launching process itself is still GPU-free."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.launcher]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _need_gpu():
    if torch.cuda.device_count() == 0:       # does not initialise HIP
        pytest.skip("no GPU")


def _run(model, dtype, tmp_path, size=256, world=2, extra_env=None, force=False, torch_ddp=False):
    from rs_detection_amd.utils import dist as rdist
    env = dict(os.environ, **(extra_env or {}))
    if torch.cuda.device_count() < world:
        env["RSDET_DIST_BACKEND"] = "gloo"
    out = str(tmp_path / "res")
    flags = "+".join((["force"] if force else []) + (["ddp"] if torch_ddp else []))
    rc, text = rdist.launch_ranks(world, [WORKER, model, dtype, out, str(size)] + ([flags] if flags else []),
                                  env=env, timeout=900)
    assert rc == 0, "a rank failed (rc %d)\n%s" % (rc, text[-2000:])
    return [json.load(open("%s.rank%d.json" % (out, r))) for r in range(world)]


@pytest.mark.timeout(1000)
@pytest.mark.parametrize("model,dtype,tol,torch_ddp", [
    ("s2anet", "f32", 1e-3, False), ("s2anet", "f32cl", 1e-3, False), ("s2anet", "bf16", 0.1, False),
    ("orcnn", "f32", 5e-3, False), ("orcnn", "bf16", 0.1, False),
    ("s2anet", "bf16", 0.1, True), ("orcnn", "f32", 5e-3, True)])          # the same harness through torch's DDP
def test_two_rank_ddp_real_model(model, dtype, tol, torch_ddp, tmp_path):
    """Two ranks of the real model: gradients after the data-parallel reduction (utils/reducer.GradReducer, or torch's
    DistributedDataParallel where ``torch_ddp``) == the mean over the shards of a single un-wrapped model's."""
    _need_gpu()
    res = _run(model, dtype, tmp_path, torch_ddp=torch_ddp)
    print(res)
    for r in res:
        assert r["world"] == 2 and r["finite"]
        # all-reduced gradients == mean of the per-shard single-process gradients, down to the run-to-run noise of the
        # single-process computation itself (measured in the worker: fp32 S2ANet ~1e-4 from MIOpen's atomics; Oriented
        # R-CNN is bimodal, 5e-8 or ~1e-3, when a tie in the proposal top-k flips one sampled RoI; bf16 ~5 %).  A DDP
        # that did not reduce (or summed instead of averaging) would sit at O(1): the shards are different images.
        assert r["grad_rel_err"] < max(tol, 3 * r["noise"]), r
        assert r["param_spread"] == 0.0, r           # bit-identical parameters on both ranks after two steps
        assert r["grad_norm"] > 0 and r["n_grad"] > 1e6
        if model == "s2anet":
            # the two train steps above ran FusedSGD on gradients that are VIEWS into the reducer's buckets; in bf16 with
            # bf16 parameters, whose gradients travel as bf16
            assert r["optimizer"] == "FusedSGD" and r["bucket_view"], r
            assert r["bf16_params"] == (dtype == "bf16"), r
        if not torch_ddp:
            assert r["reducer"] == "GradReducer" and r["grads_in_buckets"] and 1 <= r["n_buckets"] <= 8, r
            if dtype == "bf16" and model == "s2anet":
                assert "torch.bfloat16" in r["wire"], r
    assert res[0]["loss"] != res[1]["loss"]          # the ranks really worked on different shards


@pytest.mark.timeout(1000)
@pytest.mark.parametrize("model,dtype,tol", [("s2anet", "f32cl", 1e-3), ("s2anet", "bf16", 0.1), ("orcnn", "f32", 5e-3)])
def test_rccl_reducer_path_on_one_gpu(model, dtype, tol, tmp_path):
    """The backend the node will run: an ``nccl`` (= RCCL) process group of ONE rank with the gradient reducer forced on
    (``Runner(distributed="force")``): bucket views, bf16 buckets, the asynchronous all-reduces flushed from inside
    backward, FusedSGD on bucket-view gradients, ``sync_mean`` and the MIN / MAX all-reduces of the worker all go through
    RCCL on this GPU.
    With one shard the all-reduced mean IS the un-wrapped model's gradient: equal to the run-to-run noise of the
    single-process computation (reference collective: optims/optimizer.py:30-31, utils/general.py:30-48)."""
    _need_gpu()
    (r,) = _run(model, dtype, tmp_path, world=1, extra_env={"RSDET_DIST_BACKEND": "nccl"}, force=True)
    print(r)
    assert r["backend"] == "nccl" and r["world"] == 1 and r["finite"]
    assert r["bucket_view"] and r["grad_norm"] > 0 and r["n_grad"] > 1e6
    assert r["grad_rel_err"] < max(tol, 3 * r["noise"]), r
    assert r["local_rel_err"] < max(tol, 3 * r["noise"]), r
    assert r["param_spread"] == 0.0
    assert r["sync_mean"] == {"a": 1.0, "b": 3.0}
    assert r["reducer"] == "GradReducer" and r["grads_in_buckets"]
    if model == "s2anet":
        assert r["optimizer"] == "FusedSGD" and r["bf16_params"] == (dtype == "bf16")
        if dtype == "bf16":
            assert "torch.bfloat16" in r["wire"]       # bf16 on the wire


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("n", [2, 8])
def test_bench_self_launches_its_ranks(tmp_path, n):
    """`python bench.py --gpus N` (no launcher, no WORLD_SIZE): the parent starts N child ranks and relays rank 0's
    single JSON line (the driver's N>1 contract); exit code 0.  N = 8 is the driver's largest run: eight ranks on this
    box's one GPU over gloo prove the launcher / relay / barrier path at that width (not a scaling number)."""
    _need_gpu()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["RSDET_BENCH_TILE"] = "256"                  # N S2ANet replicas on one GPU: keep the tiles small
    # Eight processes time-slicing ONE GPU is not a configuration the product runs in (one process per GPU), and on this
    # pool it has a platform flake: in ~1 launch of 10 one rank's queue aborts with HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION
    # inside its very first torch kernels (faulthandler + HIP_LAUNCH_BLOCKING: the `torch.zeros` of the gradient bucket
    # right after the gloo parameter broadcast -- no kernel of this repo has run yet; the round-5 tree shows it at the same
    # rate, 2 / 20).  That one error, and only that one, is retried; anything else fails the test at once.
    for attempt in range(4):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
                            "--no-kernels", "--no-bf16-leg"], env=env, capture_output=True, text=True, timeout=1400)
        if p.returncode == 0 or "HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION" not in p.stderr:
            break
        print("attempt %d: a rank's queue aborted with HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION (platform flake), retrying" % attempt)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == n and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["global_batch"] == 4 * n and line["config"]["parallelism"] == "dp%d" % n
    assert len(lines[0]) < 8192                      # the driver's record keeps ~8 KB of the line
    assert line["cpu_baseline"] is None              # timed at N=1 only
