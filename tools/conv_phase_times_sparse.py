"""conv_phase_times.py for the SPARSE launches: where wave 0 of every workgroup of conv3d_split_kernel spends its cycles
when most items are empty or hold a few active voxels -- conv1 / conv2 of a PVConv on the chain's own clouds
(tools/scratch/chain_clouds.npz) and on the flat micro-benchmark cloud.  Needs tools/exp/liblion_timing.so
(tools/build_timing_lib.sh):   LION_HIP_SO=$PWD/tools/exp/liblion_timing.so python tools/conv_phase_times_sparse.py"""
import ctypes, os, sys, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd import _lib, fused_ops as fo
from lion_amd.functional.backend import _backend as bk
lib = _lib.load()
lib.lion_debug_split_phases.restype = ctypes.c_int
lib.lion_debug_split_phases.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.lion_debug_split_clk.restype = ctypes.c_int
lib.lion_debug_split_clk.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["item prologue", "barrier A (chunk start)", "loads + wait + activate + max", "barrier B (max)", "cut + LDS write",
         "group barrier (weights)", "taps of a group", "epilogue"]
B = 32
chain = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scratch", "chain_clouds.npz"))
def cloud(kind, n):
    if kind == "flat":
        return torch.randn(B, 3, n, device="cuda") * torch.tensor([1.0, 0.15, 0.6], device="cuda").view(1, 3, 1)
    if kind == "dense-random":
        return None
    co = torch.from_numpy(np.ascontiguousarray(chain[kind].transpose(0, 2, 1))).cuda().float()[:B]
    m = co.shape[2]
    while m > n:
        m //= 2
        co = bk.gather_features_forward(co, bk.furthest_point_sampling(co, m))
    return co.contiguous()
def phases(fn, n=5):
    buf, clk = (ctypes.c_ulonglong * 8)(), (ctypes.c_ulonglong * 2)()
    for _ in range(2): fn()
    torch.cuda.synchronize()
    lib.lion_debug_split_phases(buf, 1); lib.lion_debug_split_clk(clk, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    lib.lion_debug_split_phases(buf, 1); lib.lion_debug_split_clk(clk, 1)
    return [x / n for x in buf], a.elapsed_time(b) / n * 1e3, clk[0] / max(clk[1], 1) * 100.0
for c, r, n in ((64, 32, 2048), (128, 16, 1024)):
    conv1 = torch.nn.Conv3d(c, c, 3, padding=1).cuda(); conv2 = torch.nn.Conv3d(c, c, 3, padding=1).cuda()
    A = torch.rand(B, c, device="cuda") + 0.5; Bs = torch.randn(B, c, device="cuda") * 0.5
    for kind in ("step_0400", "step_0000", "flat"):
        co = cloud(kind, n)
        feat = torch.randn(B, c, n, device="cuda")
        out, _, _, cnt = bk.voxelize_points_forward(feat, co, r, True, 0.0)
        grid = out.view(B, c, r, r, r)
        with torch.no_grad():
            y1, _ = fo.conv3d_fused(grid, conv1, None, True, None)
            for nm, fn in (("conv1 sparse", lambda: fo.conv3d_fused(grid, conv1, None, True, fo.conv3d_occupancy(cnt, r, c, B)[0])),
                           ("conv2 delta ", lambda: fo.conv3d_fused(y1, conv2, (A, Bs), True, fo.conv3d_occupancy(cnt, r, c, B)[1], prev_conv=conv1)),
                           ("conv1 dense ", lambda: fo.conv3d_fused(grid, conv1, None, True, None))):
                ph, us, mhz = phases(fn)
                tot = sum(ph)
                print(f"C={c} r={r} {kind:9s} {nm}: {us:6.0f} us/launch (incl. occupancy launch), wave-0 cycles/launch {tot:.3e}, sclk ~{mhz:.0f} MHz | " +
                      "  ".join(f"{names[k].split(' (')[0][:18]} {100.0 * ph[k] / tot:4.1f}%" for k in range(8)), flush=True)
