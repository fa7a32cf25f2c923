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
for cin, cout, r in [(64, 64, 32), (32, 32, 32), (128, 128, 16), (128, 128, 8)]:
    x = torch.randn(B, cin, r, r, r, device="cuda"); gy = torch.randn(B, cout, r, r, r, device="cuda")
    w = torch.randn(cout, cin, 3, 3, 3, device="cuda")
    ref = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1, 1], [1, 1, 1], [1, 1, 1], False, [0, 0, 0], 1, [False, True, False])[1]
    got = conv3d_k3_wgrad(x, gy, w.shape)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    fl = 2.0 * 27 * cin * cout * r ** 3 * B
    t1 = t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1, 1], [1, 1, 1], [1, 1, 1], False, [0, 0, 0], 1, [False, True, False]))
    t2 = t(lambda: conv3d_k3_wgrad(x, gy, w.shape))
    print(f"wgrad {cin}->{cout} r={r}: rel err {err:.1e} miopen {t1*1e6:8.0f} us {fl/t1/1e12:6.1f} TF | mfma {t2*1e6:8.0f} us {fl/t2/1e12:6.1f} TF", flush=True)
