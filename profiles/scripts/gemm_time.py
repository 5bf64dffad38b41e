import torch, sys
dev = torch.device('cuda')
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
B, O, C9 = 4, 256, 2304
for H in (128, 64, 32):
    hw = H * H
    col = torch.randn(C9, B * hw, device=dev)
    go = torch.randn(B, O, hw, device=dev)
    w = torch.randn(O, C9, device=dev)
    out = torch.empty(B, O, hw, device=dev)
    gcolT = torch.empty(B * hw, C9, device=dev)
    fl = 2 * O * C9 * hw * B
    def fwd():
        for b in range(B): torch.mm(w, col[:, b * hw:(b + 1) * hw], out=out[b])
    def bdat():
        for b in range(B): torch.mm(go[b].t(), w, out=gcolT[b * hw:(b + 1) * hw])
    def bw():
        gw = torch.zeros(O, C9, device=dev)
        for b in range(B): gw.addmm_(go[b], col[:, b * hw:(b + 1) * hw].t())
        return gw
    def bw_split(S=16):
        k = hw // S; J = B * S
        go2 = go.transpose(0, 1).reshape(O, B * hw)
        parts = torch.bmm(go2.view(O, J, k).permute(1, 0, 2), col.view(C9, J, k).permute(1, 2, 0))
        return parts.sum(0)
    def bdat_one():
        go2 = go.transpose(0, 1).reshape(O, B * hw)
        torch.mm(go2.t(), w, out=gcolT)
    ref = bw(); got = bw_split()
    err = float((ref - got).abs().max() / ref.abs().max())
    line = "H=%d  fwd %7.1f us (%5.1f TF)  bwd-data %7.1f us (%5.1f TF) one-gemm %7.1f us  bwd-w loop %7.1f us (%5.1f TF)" % (
        H, t(fwd), fl / t(fwd) / 1e6, t(bdat), fl / t(bdat) / 1e6, t(bdat_one), t(bw), fl / t(bw) / 1e6)
    for S in (4, 16, 64):
        if hw % S == 0:
            tt = t(lambda: bw_split(S)); line += "  split%d %7.1f us (%5.1f TF)" % (S, tt, fl / tt / 1e6)
    print(line, " relerr %.1e" % err)
