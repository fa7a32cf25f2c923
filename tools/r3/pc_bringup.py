"""bring-up of the producer / consumer split convolution: each mode against the exact-fp32 kernel, small first"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lion_amd.conv_ops import conv3d_k3
from lion_amd import fused_ops as fo
from lion_amd.functional.backend import _backend as bk

def err(a, b): return ((a.double() - b.double()).abs().max() / b.double().abs().max()).item()

torch.manual_seed(0)
for cin, cout, r, B in [(32, 32, 16, 1), (64, 64, 32, 2), (16, 32, 16, 2), (128, 128, 16, 3), (64, 64, 32, 32)]:
    conv = torch.nn.Conv3d(cin, cout, 3, padding=1).cuda()
    x = torch.randn(B, cin, r, r, r, device="cuda")
    with torch.no_grad():
        t0 = time.time()
        y = conv3d_k3(x, conv.weight, conv.bias, split=True); torch.cuda.synchronize()
        t1 = time.time()
        y32 = conv3d_k3(x, conv.weight, conv.bias, split=False); torch.cuda.synchronize()
        print(f"dense {cin}->{cout} r{r} B{B}: err {err(y, y32):.2e}  finite {bool(torch.isfinite(y).all())}  ({t1-t0:.3f}s)", flush=True)
        A = torch.rand(B, cin, device="cuda") + 0.5; Bs = torch.randn(B, cin, device="cuda") * 0.5
        y, st = fo.conv3d_fused(x, conv, (A, Bs), True, None, split=True); torch.cuda.synchronize()
        y32, st32 = fo.conv3d_fused(x, conv, (A, Bs), True, None, split=False)
        print(f"  pro+stats: err {err(y, y32):.2e}  stats err {err(st.sum(2), st32.sum(2)):.2e}", flush=True)
# sparse plan on flat clouds
for c, r, n, B in ((64, 32, 2048, 4), (128, 16, 1024, 4), (64, 32, 2048, 32)):
    conv1 = torch.nn.Conv3d(c, c, 3, padding=1).cuda(); conv2 = torch.nn.Conv3d(c, c, 3, padding=1).cuda()
    A = torch.rand(B, c, device="cuda") + 0.5; Bs = torch.randn(B, c, device="cuda") * 0.5
    coords = torch.randn(B, 3, n, device="cuda") * torch.tensor([1, 0.15, 0.6], device="cuda").view(1, 3, 1)
    feat = torch.randn(B, c, n, device="cuda")
    out, _, _, cnt = bk.voxelize_points_forward(feat, coords, r, True, 0.0)
    grid = out.view(B, c, r, r, r)
    with torch.no_grad():
        o1, o2 = fo.conv3d_occupancy(cnt, r, c, B)
        yd, sd = fo.conv3d_fused(grid, conv1, None, True, None, split=True)
        ys, ss = fo.conv3d_fused(grid, conv1, None, True, o1, split=True); torch.cuda.synchronize()
        print(f"sparse conv1 C={c} r={r} B={B}: identical to dense {bool(torch.equal(yd, ys))}  stats identical {bool(torch.equal(sd, ss))}", flush=True)
        y2d, s2d = fo.conv3d_fused(yd, conv2, (A, Bs), True, None, split=True)
        o1, o2 = fo.conv3d_occupancy(cnt, r, c, B)
        y2s, s2s = fo.conv3d_fused(yd, conv2, (A, Bs), True, o2, prev_conv=conv1, split=True); torch.cuda.synchronize()
        print(f"  delta conv2: err vs dense {err(y2s, y2d):.2e}  stats err {err(s2s.sum(2), s2d.sum(2)):.2e}", flush=True)
# determinism
conv = torch.nn.Conv3d(64, 64, 3, padding=1).cuda(); x = torch.randn(8, 64, 32, 32, 32, device="cuda")
with torch.no_grad():
    a = conv3d_k3(x, conv.weight, conv.bias, split=True).clone()
    same = all(torch.equal(a, conv3d_k3(x, conv.weight, conv.bias, split=True)) for _ in range(10))
print("deterministic over 10 runs:", same)
