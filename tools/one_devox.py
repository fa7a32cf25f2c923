"""Driver for the profiler scripts: devoxelize (C, N, r) at B = 32; `affine` as 4th argument: AdaGN x SE scale / shift folded in (one launch);
`planned`: the two-step form a PVConv runs (plan once, planned affine forward)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.functional.backend import _backend as bk
from lion_amd import fused_ops as fo
B, C, N, r = 32, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = torch.Generator(device="cuda").manual_seed(0)
co = torch.randn(B, 3, N, device="cuda", generator=g)
_, nc, _, _ = bk.voxelize_points_forward(None, co, r, True, 0.0)
grid = torch.randn(B, C, r ** 3, device="cuda", generator=g)
if len(sys.argv) > 4 and sys.argv[4] == "planned":   # what a PVConv runs: plan once, planned affine forward per feature tensor
    sc, sh = torch.rand(B, C, device="cuda") + 0.5, torch.randn(B, C, device="cuda")
    g5 = grid.view(B, C, r, r, r)
    plan = fo.devoxelize_plan(nc, r)
    for _ in range(10): fo.devoxelize_affine(g5, nc, r, sc, sh, plan=plan)
elif len(sys.argv) > 4 and sys.argv[4] == "affine":
    sc, sh = torch.rand(B, C, device="cuda") + 0.5, torch.randn(B, C, device="cuda")
    g5 = grid.view(B, C, r, r, r)
    for _ in range(10): fo.devoxelize_affine(g5, nc, r, sc, sh)
else:
    for _ in range(10): bk.trilinear_devoxelize_forward(r, False, nc, grid)
torch.cuda.synchronize()
