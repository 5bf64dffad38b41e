"""Experiment: the bf16 S2ANet-R50 train step captured in-process as ONE hipGraph (torch.cuda.CUDAGraph), gated
by bit-equality against the eager step: loss and every gradient of replay 1 and replay 10 == eager on the same batch from
the same weights.  Prints eager vs replay ms/step and host ms/step.  No exec, no second process: capture happens inside the
running rank.  usage: graph_step.py [bf16|f32] [steps]"""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from rs_detection_amd.runner.runner import Runner

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
segment = sys.argv[3] if len(sys.argv) > 3 else "step"       # step | fwd | fwdbwd | trunk: what the graph holds
dev = torch.device("cuda", 0)
cfg = bench.s2anet_cfg()
amp = torch.bfloat16 if dt == "bf16" else None
torch.manual_seed(0)
runner = Runner(cfg, device=dev, memory_format=torch.channels_last, amp_dtype=amp, bf16_params=True)
batches = bench.make_batches(1, 4, 0, 15, dev, torch.channels_last, False)
images, targets = batches[0]
if runner.scheduler is not None:
    runner.scheduler = None                      # constant lr: the captured optimizer launch bakes its scalar arguments
for _ in range(6):
    runner.train_step(images, targets)
torch.cuda.synchronize()


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3


def state():
    return ([p.detach().clone() for p in runner.model.parameters()],
            copy.deepcopy({k: (v.clone() if torch.is_tensor(v) else v) for st in runner.optimizer.state.values() for k, v in st.items()}) if False else None)


def snapshot():
    sd = {k: v.detach().clone() for k, v in runner.model.state_dict().items()}
    opt = [{k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()} for st in runner.optimizer.state.values()]
    return sd, opt


def restore(snap):
    sd, opt = snap
    with torch.no_grad():
        for k, v in runner.model.state_dict().items():
            v.copy_(sd[k])
        for st, sv in zip(runner.optimizer.state.values(), opt):
            for k, v in st.items():
                if torch.is_tensor(v):
                    v.copy_(sv[k])
    from rs_detection_amd.ops.weight_prep import bump_epoch
    bump_epoch()


ms_eager, host_eager = timed(lambda: runner.train_step(images, targets), steps)
print("eager : %.2f ms/step, host enqueue %.2f ms/step" % (ms_eager, host_eager))

snap = snapshot()
# eager reference: 10 steps from the snapshot, loss of each and the gradients after step 1 and step 10
ref_loss, ref_grads = [], {}
for i in range(10):
    loss, _ = runner.train_step(images, targets)
    ref_loss.append(float(loss))
    if i in (0, 9):
        ref_grads[i] = [None if p.grad is None else p.grad.detach().clone() for p in runner.model.parameters()]
restore(snap)

def fwd_only():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp is not None):
        from rs_detection_amd.utils.general import parse_losses
        total, _ = parse_losses(runner.model(images, targets))
    return total


def trunk_only():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp is not None):
        feats = runner.model.neck(runner.model.backbone(images.contiguous(memory_format=torch.channels_last)))
    return sum(f.float().sum() for f in feats)


if segment != "step":
    fn = {"fwd": fwd_only, "fwdbwd": fwd_only, "trunk": trunk_only, "trunkbwd": trunk_only}[segment]
    bwd = segment.endswith("bwd")

    def run():
        runner.optimizer.zero_grad(set_to_none=True)
        l = fn()
        if bwd:
            l.backward()
        return l
    for _ in range(3):
        run()
    ms_e, host_e = timed(run, steps)
    print("segment %s eager: %.2f ms, host %.2f ms" % (segment, ms_e, host_e))
    ref = float(run())
    ref_g = [None if p.grad is None else p.grad.detach().clone() for p in runner.model.parameters()]
    g = torch.cuda.CUDAGraph()
    res = {}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            run()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    import gc
    runner.optimizer.zero_grad(set_to_none=True)
    gc.collect()                                   # (autograd graphs held by reference cycles keep their AccumulateGrad nodes --
    print("capturing segment", segment, flush=True)  #  and those nodes' streams -- alive)
    with torch.cuda.graph(g):
        res["l"] = run()
    print("capture: ok", flush=True)
    for i in range(10):
        g.replay()
    torch.cuda.synchronize()
    gr = [None if p.grad is None else p.grad for p in runner.model.parameters()]
    bad = sum(1 for a, b in zip(gr, ref_g) if (a is None) != (b is None) or (a is not None and not torch.equal(a, b)))
    print("replay 10: value %.6f (eager %.6f) %s; gradients that differ: %d of %d" % (
        float(res["l"]), ref, "==" if float(res["l"]) == ref else "!=", bad, len(gr)))
    ms_g, host_g = timed(g.replay, steps)
    print("segment %s graph: %.2f ms, host %.2f ms (eager %.2f / %.2f)" % (segment, ms_g, host_g, ms_e, host_e))
    sys.exit(0)

g = torch.cuda.CUDAGraph()
out = {}
try:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            runner.train_step(images, targets)
    torch.cuda.current_stream().wait_stream(side)
    restore(snap)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        loss, _ = runner.train_step(images, targets)
        out["loss"] = loss
    print("capture: ok")
except Exception as e:                            # noqa: BLE001
    print("capture FAILED: %s: %s" % (type(e).__name__, str(e)[:600]))
    sys.exit(0)
restore(snap)
torch.cuda.synchronize()
ok = True
for i in range(10):
    g.replay()
    torch.cuda.synchronize()
    l = float(out["loss"])
    same = l == ref_loss[i]
    if i in (0, 9):
        gr = [None if p.grad is None else p.grad for p in runner.model.parameters()]
        bad = sum(1 for a, b in zip(gr, ref_grads[i]) if (a is None) != (b is None) or (a is not None and not torch.equal(a, b)))
        print("replay %2d: loss %.6f (eager %.6f) %s; gradients that differ: %d of %d" % (i + 1, l, ref_loss[i], "==" if same else "!=", bad, len(gr)))
        ok &= bad == 0
    ok &= same
print("bit-equality over 10 replays:", ok)
ms_g, host_g = timed(g.replay, steps)
print("graph : %.2f ms/step, host %.2f ms/step (eager %.2f / %.2f)" % (ms_g, host_g, ms_eager, host_eager))
