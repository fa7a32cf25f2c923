"""dense vs sparse evaluation of the two convolutions of a PVConv voxel branch (B=32) on synthetic clouds -- Gaussian
(the chain's start), flat (airplane-like), clumped (95 % of the points in a tenth of the extent: the chain after ~20 steps
with random-init weights) -- and, when tools/scratch/chain_clouds.npz exists (tools/dump_chain_clouds.py), on the x_t of
the real chain.  Reports the fraction of empty tiles, of ACTIVE voxels (a point within the margin: what the round-5
voxel compaction computes) and the MFMA column blocks per tile relative to dense; the sparse times include the occupancy
launch (its own time is printed)."""
import sys, os, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
chain = None
cf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scratch", "chain_clouds.npz")
if os.path.exists(cf):
    chain = np.load(cf)
def clouds(n):
    g = torch.Generator(device="cuda").manual_seed(0)
    out = []
    for name, sc in (("gauss", [1, 1, 1]), ("flat", [1, 0.15, 0.6])):
        out.append((name, torch.randn(B, 3, n, device="cuda", generator=g) * torch.tensor(sc, device="cuda", dtype=torch.float32).view(1, 3, 1)))
    cl = torch.randn(B, 3, n, device="cuda", generator=g)
    cl[:, :, : int(0.95 * n)] *= 0.1
    out.append(("clump", cl))
    if chain is not None:
        for k in ("step_0000", "step_0020", "step_0400"):
            co = torch.from_numpy(np.ascontiguousarray(chain[k].transpose(0, 2, 1))).cuda().float()[:B]
            m = co.shape[2]
            while m > n:       # the r = 16 grid sees the FPS subset of the step's cloud
                m //= 2
                idx = bk.furthest_point_sampling(co, m)
                co = bk.gather_features_forward(co, idx)
            out.append((k[5:], co.contiguous()))
    return out
for c, r, n in ((64, 32, 2048), (32, 32, 2048), (128, 16, 1024)):
    conv1 = torch.nn.Conv3d(c, c, 3, padding=1).cuda(); conv2 = torch.nn.Conv3d(c, c, 3, padding=1).cuda()
    A = torch.rand(B, c, device="cuda") + 0.5; Bs = torch.randn(B, c, device="cuda") * 0.5
    for name, coords in clouds(n):
        feat = torch.randn(B, c, n, device="cuda")
        out, _, _, cnt = bk.voxelize_points_forward(feat, coords, r, True, 0.0)
        grid = out.view(B, c, r, r, r)
        nt = lib.lion_conv3d_stat_tiles(r, c, B, 1)
        o1, o2 = fo.conv3d_occupancy(cnt, r, c, B)
        fr = [1 - ((o.view(-1)[:B * nt] & 0xf) != 0).float().mean().item() for o in (o1, o2)]
        # active voxels (a point within the margin) from the count grid itself (round 5's bit maps left the occupancy buffer)
        g_ = (cnt.view(B, 1, r, r, r) > 0).float()
        act = [torch.nn.functional.max_pool3d(g_, 2 * m_ + 1, 1, m_).mean().item() for m_ in (1, 2)]
        blk = [float("nan"), float("nan")]
        t_occ = t(lambda: fo.conv3d_occupancy(cnt, r, c, B))
        with torch.no_grad():
            y1, _ = fo.conv3d_fused(grid, conv1, None, True, None)
            print(f"C={c} r={r} {name:5s} empty tiles m1 {fr[0]:.2f} m2 {fr[1]:.2f} active voxels {act[0]:.3f} {act[1]:.3f} blocks {blk[0]:.3f} {blk[1]:.3f} | conv1 dense "
                  f"{t(lambda: fo.conv3d_fused(grid, conv1, None, True, None)):6.0f} sparse "
                  f"{t(lambda: fo.conv3d_fused(grid, conv1, None, True, fo.conv3d_occupancy(cnt, r, c, B)[0])) - t_occ:6.0f} | conv2 dense "
                  f"{t(lambda: fo.conv3d_fused(y1, conv2, (A, Bs), True, None)):6.0f} delta "
                  f"{t(lambda: fo.conv3d_fused(y1, conv2, (A, Bs), True, fo.conv3d_occupancy(cnt, r, c, B)[1], prev_conv=conv1)) - t_occ:6.0f} us  (occupancy launch {t_occ:.0f} us)", flush=True)
