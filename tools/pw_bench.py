"""fp32-MFMA vs split-operand 1x1 convolution (csrc/pwconv.hip / pwconv_split.hip) at shapes of the local denoiser, B=32:
time per call, algorithmic GB/s (one read of x + one write of y), max error of both against float64 on one batch entry."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd import fused_ops as fo

def t_us(f, n=20):
    """n calls captured in one hipGraph, replayed: kernel time without the python between launches"""
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)

B = 32
print(f"{'Cin->Cout  L':24s} {'pro':>3s} {'fp32 us':>8s} {'GB/s':>6s} {'split us':>8s} {'GB/s':>6s} {'err32':>8s} {'errS':>8s}")
for cin, cout, L, pro in [(192, 128, 2048, False), (128, 128, 2048, True), (128, 64, 2048, True), (64, 64, 2048, False),
                          (35, 32, 32768, False), (32, 64, 32768, True), (67, 64, 8192, False), (64, 128, 8192, True),
                          (131, 128, 2048, False), (128, 256, 2048, True), (320, 256, 512, False), (256, 256, 512, True),
                          (384, 256, 128, False), (256, 128, 128, True), (128, 128, 16, False)]:
    torch.manual_seed(0)
    conv = torch.nn.Conv1d(cin, cout, 1).cuda()
    x = torch.randn(B, cin, L, device="cuda")
    A = torch.randn(B, cin, device="cuda") * 0.5 + 1.0
    Bs = torch.randn(B, cin, device="cuda") * 0.3
    p = (A, Bs) if pro else None
    with torch.no_grad():
        t32 = t_us(lambda: fo.pwconv_fused(x, conv, p, split=False))
        ts = t_us(lambda: fo.pwconv_fused(x, conv, p, split=True))
        xin = x[:1].double()
        if pro:
            t = xin * A[:1].double()[:, :, None] + Bs[:1].double()[:, :, None]
            xin = t * torch.sigmoid(t)
        ref = torch.einsum("oc,bcl->bol", conv.weight.double()[:, :, 0], xin) + conv.bias.double()[None, :, None]
        e32 = ((fo.pwconv_fused(x, conv, p, split=False)[0][:1].double() - ref).abs().max() / ref.abs().max()).item()
        es = ((fo.pwconv_fused(x, conv, p, split=True)[0][:1].double() - ref).abs().max() / ref.abs().max()).item()
    gb = B * (cin + cout) * L * 4 / 1e3
    print(f"{cin:4d}->{cout:4d} L={L:6d}      {int(pro):3d} {t32:8.1f} {gb / t32:6.0f} {ts:8.1f} {gb / ts:6.0f} {e32:8.1e} {es:8.1e}")
