"""Does the time of the split convolution depend on how long the GPU has been busy?  300 back-to-back launches of
Conv3d 64->64 @32^3 B=32, one event pair per block of 10: us per launch over time (clock ramp / power management)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.conv_ops import conv3d_k3
from lion_amd import fused_ops as fo
B, c, r = 32, 64, 32
conv = torch.nn.Conv3d(c, c, 3, padding=1).cuda()
x = torch.randn(B, c, r, r, r, device="cuda")
xz = torch.zeros_like(x)
A = torch.rand(B, c, device="cuda") + 0.5; Bs = torch.randn(B, c, device="cuda") * 0.5
def series(label, fn, blocks=30, per=10, idle=0.0):
    with torch.no_grad():
        fn(); torch.cuda.synchronize()
        if idle: time.sleep(idle)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
        ev[0].record()
        for k in range(blocks):
            for _ in range(per): fn()
            ev[k + 1].record()
        torch.cuda.synchronize()
    ts = [ev[k].elapsed_time(ev[k + 1]) / per * 1e3 for k in range(blocks)]
    print(f"{label:46s} us/launch per block of {per}: " + " ".join(f"{t:.0f}" for t in ts), flush=True)
series("plain random input (after 0.5 s idle)", lambda: conv3d_k3(x, conv.weight, conv.bias, split=True), idle=0.5)
series("plain random input (no idle)", lambda: conv3d_k3(x, conv.weight, conv.bias, split=True))
series("plain ZERO input (data toggling)", lambda: conv3d_k3(xz, conv.weight, conv.bias, split=True))
series("conv2 form (pro+stats) dense", lambda: fo.conv3d_fused(x, conv, (A, Bs), True, None))
series("exact fp32 kernel", lambda: conv3d_k3(x, conv.weight, conv.bias, split=False), blocks=10)
