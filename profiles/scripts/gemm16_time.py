import torch
dev = torch.device('cuda')
def t(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
B, O, C9 = 4, 256, 2304
for dt in (torch.float32, torch.bfloat16):
  for H in (128, 64, 32):
    hw = H * H
    col = torch.randn(C9, B * hw, device=dev).to(dt)
    go = torch.randn(B, O, hw, device=dev).to(dt)
    w = torch.randn(O, C9, device=dev).to(dt)
    out = torch.empty(B, O, hw, device=dev, dtype=dt)
    go2 = go.transpose(0, 1).reshape(O, B * hw)
    def fwd():
        for b in range(B): torch.mm(w, col[:, b * hw:(b + 1) * hw], out=out[b])
    def fwd_one():
        return torch.mm(w, col)
    def bdat_one():
        return torch.mm(go2.t(), w)
    def bw_split(S=16):
        n = B * hw; k = n // S
        return torch.bmm(go2.view(O, S, k).permute(1, 0, 2), col.view(C9, S, k).permute(1, 2, 0)).sum(0)
    def bw_one():
        return torch.mm(go2, col.t())
    print(str(dt)[6:], "H=%3d  fwd loop %7.1f  fwd one(+permute needed) %7.1f  bwd-data %7.1f  bwd-w split16 %7.1f  bwd-w one %7.1f us" % (
        H, t(fwd), t(fwd_one), t(bdat_one), t(bw_split), t(bw_one)))
