import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lion_amd import fused_ops as fo
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
B = 32
for cin, cout, r in [(64, 64, 32), (128, 128, 16), (32, 32, 32)]:
    conv = torch.nn.Conv3d(cin, cout, 3, padding=1).cuda()
    x = torch.randn(B, cin, r, r, r, device="cuda")
    A = torch.rand(B, cin, device="cuda") + 0.5; Bs = torch.randn(B, cin, device="cuda") * 0.5
    with torch.no_grad():
        print(f"{cin}->{cout} r{r}: plain {t(lambda: fo.conv3d_fused(x, conv, None, False, None, split=True)):7.1f}  stats {t(lambda: fo.conv3d_fused(x, conv, None, True, None, split=True)):7.1f}  pro+stats {t(lambda: fo.conv3d_fused(x, conv, (A, Bs), True, None, split=True)):7.1f} us", flush=True)
