import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.conv_ops import conv3d_k3
B = 32
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / it * 1e-3
out = []
for cin, cout, r in [(64, 64, 32), (32, 32, 32), (128, 128, 16), (128, 128, 8)]:
    conv = torch.nn.Conv3d(cin, cout, 3, padding=1).cuda(); x = torch.randn(B, cin, r, r, r, device="cuda")
    with torch.no_grad():
        ref = conv(x); got = conv3d_k3(x, conv.weight, conv.bias); err = (ref - got).abs().max().item() / ref.abs().max().item()
        fl = 2.0 * 27 * cin * cout * r ** 3 * B; t2 = t(lambda: conv3d_k3(x, conv.weight, conv.bias))
    out.append(f"{cin}->{cout}@{r}: {fl/t2/1e12:6.1f} TF (err {err:.1e})")
print(os.environ.get("LION_HIP_SO", "default"), " | ".join(out), flush=True)
