"""dense vs sparse evaluation of the two convolutions of a PVConv voxel branch (B=32) on synthetic clouds."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lion_amd import fused_ops as fo, _lib as L
from lion_amd.functional.backend import _backend as bk
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / it * 1e3
B = 32
lib = L.load()
for c, r, n in ((64, 32, 2048), (32, 32, 2048), (128, 16, 1024)):
    conv1 = torch.nn.Conv3d(c, c, 3, padding=1).cuda(); conv2 = torch.nn.Conv3d(c, c, 3, padding=1).cuda()
    A = torch.rand(B, c, device="cuda") + 0.5; Bs = torch.randn(B, c, device="cuda") * 0.5
    for name, sc in (("gauss", [1, 1, 1]), ("flat", [1, 0.15, 0.6])):
        coords = torch.randn(B, 3, n, device="cuda") * torch.tensor(sc, device="cuda", dtype=torch.float32).view(1, 3, 1)
        feat = torch.randn(B, c, n, device="cuda")
        out, _, _, cnt = bk.voxelize_points_forward(feat, coords, r, True, 0.0)
        grid = out.view(B, c, r, r, r)
        nt = lib.lion_conv3d_stat_tiles(r, c, B, 1)
        o1, o2 = fo.conv3d_occupancy(cnt, r, c, B)
        fr = [1 - (o1[:B * nt] != 0).float().mean().item(), 1 - (o2[:B * nt] != 0).float().mean().item()]
        with torch.no_grad():
            y1, _ = fo.conv3d_fused(grid, conv1, None, True, None)
            print("start", c, r, name, flush=True); print(f"C={c} r={r} {name:5s} empty tiles m1 {fr[0]:.2f} m2 {fr[1]:.2f} | conv1 dense "
                  f"{t(lambda: fo.conv3d_fused(grid, conv1, None, True, None)):6.0f} sparse "
                  f"{t(lambda: fo.conv3d_fused(grid, conv1, None, True, fo.conv3d_occupancy(cnt, r, c, B)[0])):6.0f} | conv2 dense "
                  f"{t(lambda: fo.conv3d_fused(y1, conv2, (A, Bs), True, None)):6.0f} delta "
                  f"{t(lambda: fo.conv3d_fused(y1, conv2, (A, Bs), True, fo.conv3d_occupancy(cnt, r, c, B)[1], prev_conv=conv1)):6.0f} us", flush=True)
