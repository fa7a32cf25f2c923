import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lion_amd.functional.backend import _backend as bk
torch.manual_seed(0)
B = 32
def frac(cnt, r, td, th, tw):
    g = (cnt.view(B, r, r, r) > 0).float()
    # halo-dilated occupancy: a tile is non-empty if any voxel within [-1,+1] of the tile is occupied
    d = torch.nn.functional.max_pool3d(g[:, None], 3, 1, 1)[:, 0]
    t = d.view(B, r // td, td, r // th, th, r // tw, tw).amax(dim=(2, 4, 6))
    return 1.0 - t.mean().item()
for name, co in (("gauss", torch.randn(B, 3, 2048, device="cuda")),
                 ("sphere-surface", torch.nn.functional.normalize(torch.randn(B, 3, 2048, device="cuda"), dim=1)),
                 ("flat (airplane-like)", torch.randn(B, 3, 2048, device="cuda") * torch.tensor([1.0, 0.15, 0.6], device="cuda").view(1, 3, 1))):
    for r, n, tiles in ((32, 2048, [(2, 4, 32), (4, 4, 32)]), (16, 1024, [(4, 4, 16), (8, 4, 16)]), (8, 256, [(2, 8, 8)])):
        c = co[:, :, :n].contiguous()
        _, nc, ind, cnt = bk.voxelize_points_forward(torch.randn(B, 4, n, device="cuda"), c, r, True, 0.0)
        print(name, "r", r, "occupied voxels %.3f" % (cnt > 0).float().mean().item(),
              " empty tiles:", {t: round(frac(cnt, r, *t), 3) for t in tiles})
