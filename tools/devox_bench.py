"""r = 32 devoxelize (the ring kernel) on Gaussian and flat clouds: min / median of 20 graph-free timings, algorithmic GB/s
(8 corner reads + 1 write per point and channel).  LION_DEVOX_RING = ring depth cap (2 = the round-3 schedule)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd.functional.backend import _backend as bk
from lion_amd import fused_ops as fo
B, N, r = 32, 2048, 32
g = torch.Generator(device="cuda").manual_seed(0)
def t1(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(it):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); fn(); fn(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 4 * 1e3)
    ts.sort(); return ts[0], ts[len(ts) // 2]
print(f"LION_DEVOX_RING={os.environ.get('LION_DEVOX_RING', '(default 4)')}")
for C in (64, 32):
    for name, sc in (("gauss", [1, 1, 1]), ("flat", [1, 0.15, 0.6])):
        co = torch.randn(B, 3, N, device="cuda", generator=g) * torch.tensor(sc, device="cuda").view(1, 3, 1)
        _, nc, _, _ = bk.voxelize_points_forward(None, co, r, True, 0.0)
        grid = torch.randn(B, C, r ** 3, device="cuda", generator=g)
        scl, shf = torch.rand(B, C, device="cuda") + 0.5, torch.randn(B, C, device="cuda")
        g5 = grid.view(B, C, r, r, r)
        nbytes = 4 * B * (3 * N + C * 8 * N + C * N)
        plan = fo.devoxelize_plan(nc, r)
        for label, fn in (("affine", lambda: fo.devoxelize_affine(g5, nc, r, scl, shf)), ("planned", lambda: fo.devoxelize_affine(g5, nc, r, scl, shf, plan=plan)),
                          ("plan", lambda: fo.devoxelize_plan(nc, r))):
            mn, md = t1(fn)
            print(f"C={C} {name:5s} {label:6s} min {mn:6.1f} us  median {md:6.1f} us  -> {nbytes / md / 1e3:6.0f} GB/s = {nbytes / md / 1e3 / 8000:.3f} of 8 TB/s (min: {nbytes / mn / 1e3 / 8000:.3f})", flush=True)
