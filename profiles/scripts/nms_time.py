import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rs_detection_amd import ops
from rs_detection_amd.ops.nms_rotated import _label_major_order
from rs_detection_amd.utils import synthetic as syn
dev=torch.device('cuda')
def gtime(f):
    for _ in range(3): k = f()
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): f()
    g.replay(); torch.cuda.synchronize()
    st=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    st.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return st.elapsed_time(e)*100, k
def run(M):
    d, s, l = syn.nms_cluster_boxes(M)
    d6 = torch.from_numpy(np.concatenate([d, l[:, None].astype(np.float32)], 1)).to(dev)
    sc = torch.from_numpy(s).to(dev); lab = torch.from_numpy(l).to(dev)
    order = torch.argsort(sc, descending=True, stable=True).int()
    lorder = _label_major_order(sc, lab).int()
    t1,k1 = gtime(lambda: ops.nms_rotated_keep_mask(d6, order, 0.1, 6))
    t2,k2 = gtime(lambda: ops.nms_rotated_keep_mask(d6, lorder, 0.1, 6, label_major=True))
    t3,k3 = gtime(lambda: ops.nms_rotated_keep_mask(d6[:, :5].contiguous(), order, 0.1, 5))
    assert torch.equal(k1, k2)
    print("M=%d: score order %.1f us | label-major order %.1f us (same keep, %d kept) | single class %.1f us (%d kept)"%(M, t1, t2, int(k1.sum()), t3, int(k3.sum())))
for M in ([int(x) for x in sys.argv[1:]] or (1000, 5344, 20000)): run(M)
