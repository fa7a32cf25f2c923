import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.functional.backend import _backend as bk
B, C, N, r = 32, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g = torch.Generator(device="cuda").manual_seed(0)
co = torch.randn(B, 3, N, device="cuda", generator=g)
_, nc, _, _ = bk.voxelize_points_forward(None, co, r, True, 0.0)
grid = torch.randn(B, C, r ** 3, device="cuda", generator=g)
for _ in range(10): bk.trilinear_devoxelize_forward(r, False, nc, grid)
torch.cuda.synchronize()
