"""five launches of the Conv3d weight-gradient kernel at one shape (B = 32), for rocprofv3 counter passes.
usage: one_wgrad.py CIN COUT R"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.conv_ops import conv3d_k3_wgrad
cin, cout, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(32, cin, r, r, r, device="cuda"); gy = torch.randn(32, cout, r, r, r, device="cuda")
for _ in range(5): conv3d_k3_wgrad(x, gy, (cout, cin, 3, 3, 3))
torch.cuda.synchronize()
