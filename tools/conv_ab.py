"""A/B of one library build on the voxel convolutions of a PVConv (B = 32), every launch inside ONE hipGraph replay
(the kernels' own time): dense forms (plain, conv1 form = statistics, conv2 form = prologue + statistics) and the sparse
forms (occupancy + conv1, occupancy + constant response + delta conv2) on Gaussian / flat / clumped clouds and on the
chain's own x_t (tools/scratch/chain_clouds.npz).  Run once per library:
    LION_HIP_SO=$PWD/tools/scratch/liblion_hip_r04.so python tools/conv_ab.py r04
    python tools/conv_ab.py r05
The sparse times INCLUDE the occupancy launch (and conv2's constant-response launch): what a PVConv pays."""
import os, sys, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd import fused_ops as fo
from lion_amd.functional.backend import _backend as bk
tag = sys.argv[1] if len(sys.argv) > 1 else "lib"
B = 32
def tg(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(it): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
chain = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scratch", "chain_clouds.npz"))
def clouds(n):
    g = torch.Generator(device="cuda").manual_seed(0)
    out = [("gauss", torch.randn(B, 3, n, device="cuda", generator=g)),
           ("flat", torch.randn(B, 3, n, device="cuda", generator=g) * torch.tensor([1.0, 0.15, 0.6], device="cuda").view(1, 3, 1))]
    for k in ("step_0000", "step_0020", "step_0400"):
        co = torch.from_numpy(np.ascontiguousarray(chain[k].transpose(0, 2, 1))).cuda().float()[:B]
        m = co.shape[2]
        while m > n:
            m //= 2
            co = bk.gather_features_forward(co, bk.furthest_point_sampling(co, m))
        out.append((k[5:], co.contiguous()))
    return out
with torch.no_grad():
    for c, r, n in ((64, 32, 2048), (32, 32, 2048), (128, 16, 1024)):
        conv1 = torch.nn.Conv3d(c, c, 3, padding=1).cuda(); conv2 = torch.nn.Conv3d(c, c, 3, padding=1).cuda()
        A = torch.rand(B, c, device="cuda") + 0.5; Bs = torch.randn(B, c, device="cuda") * 0.5
        xr = torch.randn(B, c, r, r, r, device="cuda")
        print(f"{tag} C={c} r={r} random dense operands: plain {tg(lambda: fo.conv3d_fused(xr, conv1, None, False, None)):6.0f}  "
              f"conv1 form {tg(lambda: fo.conv3d_fused(xr, conv1, None, True, None)):6.0f}  "
              f"conv2 form {tg(lambda: fo.conv3d_fused(xr, conv2, (A, Bs), True, None)):6.0f} us", flush=True)
        for name, coords in clouds(n):
            feat = torch.randn(B, c, n, device="cuda")
            out, _, _, cnt = bk.voxelize_points_forward(feat, coords, r, True, 0.0)
            grid = out.view(B, c, r, r, r)
            y1, _ = fo.conv3d_fused(grid, conv1, None, True, None)
            t_occ = tg(lambda: fo.conv3d_occupancy(cnt, r, c, B))
            print(f"{tag} C={c} r={r} {name:5s} | conv1 dense {tg(lambda: fo.conv3d_fused(grid, conv1, None, True, None)):6.0f} "
                  f"sparse {tg(lambda: fo.conv3d_fused(grid, conv1, None, True, fo.conv3d_occupancy(cnt, r, c, B)[0])):6.0f} | conv2 dense "
                  f"{tg(lambda: fo.conv3d_fused(y1, conv2, (A, Bs), True, None)):6.0f} delta "
                  f"{tg(lambda: fo.conv3d_fused(y1, conv2, (A, Bs), True, fo.conv3d_occupancy(cnt, r, c, B)[1], prev_conv=conv1)):6.0f} us  "
                  f"(occupancy launch alone {t_occ:.0f} us)", flush=True)
