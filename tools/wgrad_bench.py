"""Conv3d weight gradient: exact-fp32 MFMA kernel vs the split-operand kernel (round 4) at the layer shapes of a training
step, B = 32; error of both against float64 on a B = 2 slice; --miopen adds the library's backward-filter."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.conv_ops import conv3d_k3_wgrad
def t(fn, it=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / it * 1e-3
B = 32
for cin, cout, r in [(64, 64, 32), (32, 32, 32), (128, 128, 16), (128, 128, 8), (192, 128, 8)]:
    x = torch.randn(B, cin, r, r, r, device="cuda"); gy = torch.randn(B, cout, r, r, r, device="cuda")
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda")
    xd = x[:2].double().requires_grad_(False); wd = w.double().requires_grad_(True)
    torch.nn.functional.conv3d(xd, wd, None, padding=1).backward(gy[:2].double())
    e32 = ((conv3d_k3_wgrad(x[:2].contiguous(), gy[:2].contiguous(), w.shape, split=False).double() - wd.grad).abs().max() / wd.grad.abs().max()).item()
    es = ((conv3d_k3_wgrad(x[:2].contiguous(), gy[:2].contiguous(), w.shape, split=True).double() - wd.grad).abs().max() / wd.grad.abs().max()).item()
    fl = 2.0 * 27 * cin * cout * r ** 3 * B
    t2 = t(lambda: conv3d_k3_wgrad(x, gy, w.shape, split=False))
    t3 = t(lambda: conv3d_k3_wgrad(x, gy, w.shape, split=True))
    line = f"wgrad {cin}->{cout} r={r}: fp32 {t2*1e6:8.0f} us {fl/t2/1e12:6.1f} TF (err {e32:.1e}) | split {t3*1e6:8.0f} us {fl/t3/1e12:6.1f} TF-eq (err {es:.1e})"
    if "--miopen" in sys.argv:
        t1 = t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1, 1], [1, 1, 1], [1, 1, 1], False, [0, 0, 0], 1, [False, True, False]))
        line += f" | miopen {t1*1e6:8.0f} us"
    print(line, flush=True)
