"""times the voxel convolutions of one local-prior forward (B=32) on both kernels: exact-fp32 MFMA (csrc/conv3d.hip) and
split operands (csrc/conv3d_split.hip), plain dense call, HIP events.  usage: python tools/conv_split_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.conv_ops import conv3d_k3

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n

B = 32
print(f"{'shape':24s} {'fp32 us':>9s} {'TF':>6s} {'split us':>9s} {'TF-eq':>6s}")
for cin, cout, r in [(64, 64, 32), (32, 32, 32), (128, 128, 16), (64, 128, 16), (128, 128, 8), (192, 128, 8), (256, 128, 8)]:
    conv = torch.nn.Conv3d(cin, cout, 3, padding=1).cuda()
    x = torch.randn(B, cin, r, r, r, device="cuda")
    fl = 2.0 * 27 * cin * cout * r ** 3 * B
    with torch.no_grad():
        t0 = t(lambda: conv3d_k3(x, conv.weight, conv.bias, split=False))
        t1 = t(lambda: conv3d_k3(x, conv.weight, conv.bias, split=True))
    print(f"{cin:4d}->{cout:4d} r{r:2d} B{B}        {t0:9.1f} {fl/t0/1e6:6.1f} {t1:9.1f} {fl/t1/1e6:6.1f}")
