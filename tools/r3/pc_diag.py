import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lion_amd.conv_ops import conv3d_k3
import torch.nn.functional as F
torch.manual_seed(0)
for cin, cout, r, B in [(32, 32, 16, 1), (16, 32, 16, 1), (32, 32, 32, 1), (64, 32, 16, 2)]:
    conv = torch.nn.Conv3d(cin, cout, 3, padding=1).cuda()
    x = torch.randn(B, cin, r, r, r, device="cuda")
    with torch.no_grad():
        ref = F.conv3d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
        y = conv3d_k3(x, conv.weight, conv.bias, split=True)
        y2 = conv3d_k3(x, conv.weight, conv.bias, split=True)
    d = (y.double() - ref).abs() / ref.abs().max()
    print(f"{cin}->{cout} r{r} B{B}: max {d.max().item():.2e} same twice {bool(torch.equal(y, y2))}")
    print("  per channel max:", " ".join(f"{v:.0e}" for v in d.amax(dim=(0, 2, 3, 4)).tolist()))
    dd = d.amax(dim=(0, 1))  # [d,h,w]
    print("  per d-plane:", " ".join(f"{v:.0e}" for v in dd.amax(dim=(1, 2)).tolist()))
    print("  per h:", " ".join(f"{v:.0e}" for v in dd.amax(dim=(0, 2)).tolist()))
    print("  per w:", " ".join(f"{v:.0e}" for v in dd.amax(dim=(0, 1)).tolist()))
    bad = (d > 5e-6).sum().item(); print("  count > 5e-6:", bad, "of", d.numel())
