import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rs_detection_amd import ops
from rs_detection_amd.utils import synthetic as syn
dev=torch.device('cuda')
def run(M):
    d, s, l = syn.nms_cluster_boxes(M)
    d6 = torch.from_numpy(np.concatenate([d, l[:, None].astype(np.float32)], 1)).to(dev)
    order = torch.from_numpy(np.argsort(-s, kind="stable").astype(np.int32)).to(dev)
    f = lambda: ops.nms_rotated_keep_mask(d6, order, 0.1, 6)
    for _ in range(3): k = f()
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): f()
    g.replay(); torch.cuda.synchronize()
    st=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    st.record(); g.replay(); e.record(); torch.cuda.synchronize()
    print("M=%d: %.1f us per call, kept %d"%(M, st.elapsed_time(e)*100, int(k.sum())))
for M in ([int(x) for x in sys.argv[1:]] or (1000, 5344, 20000)): run(M)
