import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.conv_ops import conv3d_k3  # LION_CONV_SPLIT=0 selects the exact-fp32 kernel
cin, cout, r = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
conv = torch.nn.Conv3d(cin, cout, 3, padding=1).cuda(); x = torch.randn(32, cin, r, r, r, device="cuda")
with torch.no_grad():
    for _ in range(5): conv3d_k3(x, conv.weight, conv.bias)
torch.cuda.synchronize()
