import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lion_amd import fused_ops as fo
from lion_amd.functional.backend import _backend as bk
c, r, n, B, flat, which = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
torch.manual_seed(0)
conv1 = torch.nn.Conv3d(c, c, 3, padding=1).cuda(); conv2 = torch.nn.Conv3d(c, c, 3, padding=1).cuda()
A = torch.rand(B, c, device="cuda") + 0.5; Bs = torch.randn(B, c, device="cuda") * 0.5
sc = [1, 0.15, 0.6] if flat else [1, 1, 1]
coords = torch.randn(B, 3, n, device="cuda") * torch.tensor(sc, device="cuda", dtype=torch.float32).view(1, 3, 1)
feat = torch.randn(B, c, n, device="cuda")
out, _, _, cnt = bk.voxelize_points_forward(feat, coords, r, True, 0.0)
grid = out.view(B, c, r, r, r)
with torch.no_grad():
    yd, _ = fo.conv3d_fused(grid, conv1, None, True, None, split=True)
    torch.cuda.synchronize()
    for it in range(5):
        o1, o2 = fo.conv3d_occupancy(cnt, r, c, B)
        t0 = time.time()
        if which == "conv1":
            y, s = fo.conv3d_fused(grid, conv1, None, True, o1, split=True)
        else:
            y, s = fo.conv3d_fused(yd, conv2, (A, Bs), True, o2, prev_conv=conv1, split=True)
        torch.cuda.synchronize()
        print(f"{which} C={c} r={r} B={B} flat={flat} iter {it}: {1e6*(time.time()-t0):.0f} us", flush=True)
