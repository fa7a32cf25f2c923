"""Driver for the profiler scripts: the conv2 form a sampling step runs (AdaGN + Swish prologue, constant + delta, GroupNorm
sums, work queue) at 64 -> 64 @ 32^3, B = 32, with every tile occupied; `conv1` as argument selects the conv1 form."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd import fused_ops as fo
B, c, r = 32, 64, 32
conv = torch.nn.Conv3d(c, c, 3, padding=1).cuda()
x = torch.randn(B, c, r, r, r, device="cuda")
ones = torch.ones(B, r ** 3, device="cuda", dtype=torch.int32)
pa, pb = torch.rand(B, c, device="cuda") + 0.5, torch.randn(B, c, device="cuda") * 0.5
with torch.no_grad():
    for _ in range(6):
        o1, o2 = fo.conv3d_occupancy(ones, r, c, B)
        if len(sys.argv) > 1 and sys.argv[1] == "conv1":
            fo.conv3d_fused(x, conv, None, True, o1)
        else:
            fo.conv3d_fused(x, conv, (pa, pb), True, o2, prev_conv=conv)
torch.cuda.synchronize()
