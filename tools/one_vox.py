"""Driver for the profiler scripts: voxelize (C, N, r) at B = 32; `scatter` as 4th argument runs the plan form
(lion_voxel_index once, lion_voxel_scatter 10 times), otherwise the fused single call."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.functional.backend import _backend as bk
B, C, N, r = 32, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = torch.Generator(device="cuda").manual_seed(0)
co = torch.randn(B, 3, N, device="cuda", generator=g); feat = torch.randn(B, C, N, device="cuda", generator=g)
if len(sys.argv) > 4 and sys.argv[4] == "scatter":
    plan = bk.voxel_index(co, r, True, 0.0)
    for _ in range(10): bk.voxel_scatter(feat, plan)
else:
    for _ in range(10): bk.voxelize_points_forward(feat, co, r, True, 0.0)
torch.cuda.synchronize()
