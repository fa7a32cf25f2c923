"""Phase cycles per item and the duration histogram of the tap groups (cycles per MFMA and wave between two group barriers)
of conv3d_split_kernel: plain kernel and the in-step forms.  Needs tools/exp/liblion_timing.so (tools/build_timing_lib.sh):
   LION_HIP_SO=$PWD/tools/exp/liblion_timing.so python tools/conv_phase_hist.py
A lone wave issues the tap pattern at 32.5 cycles per MFMA, two tapping waves on one SIMD at 65 each
(tools/exp/mfma_issue_probe.hip): the histogram says how often the two co-resident workgroups tap at the same time."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd import _lib, fused_ops as fo
from lion_amd.conv_ops import conv3d_k3
from lion_amd.functional.backend import _backend as bk
lib = _lib.load()
for f in (lib.lion_debug_split_phases, lib.lion_debug_split_hist, lib.lion_debug_split_clk):
    f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["item prologue", "barrier A (chunk start)", "loads+wait+activate+max", "barrier B (max)", "cut + LDS write",
         "group barrier (weights)", "taps", "epilogue"]
edges = ["<36", "36-42", "42-50", "50-58", "58-66", "66-76", "76-95", ">=95"]
def measure(label, fn, items, n=5):
    with torch.no_grad():
        for _ in range(3): fn()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)(); h = (ctypes.c_ulonglong * 8)()
        lib.lion_debug_split_phases(buf, 1); lib.lion_debug_split_hist(h, 1); ck = (ctypes.c_ulonglong * 2)(); lib.lion_debug_split_clk(ck, 1)
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        lib.lion_debug_split_phases(buf, 1); lib.lion_debug_split_hist(h, 1); lib.lion_debug_split_clk(ck, 1)
    tot = sum(buf); ht = max(sum(h), 1)
    print(f"{label}: {a.elapsed_time(b) / n * 1e3:.0f} us (instrumented), wave-0 cycles per item {tot / n / items:.0f}; s_memtime ticks per us of workgroup lifetime {100.0 * ck[0] / max(ck[1], 1):.0f}, workgroup lifetimes sum {ck[1] / 100.0 / n:.0f} us per launch")
    print("   " + " | ".join(f"{names[k]} {buf[k] / n / items:.0f}" for k in range(8)))
    print("   tap groups by cycles/MFMA: " + " | ".join(f"{edges[k]} {100.0 * h[k] / ht:.1f}%" for k in range(8)), flush=True)
B = 32
for c, r, n_pts in ((64, 32, 2048), (128, 16, 1024))[:int(os.environ.get('PHASE_HIST_SHAPES', '2'))]:
    conv1 = torch.nn.Conv3d(c, c, 3, padding=1).cuda(); conv2 = torch.nn.Conv3d(c, c, 3, padding=1).cuda()
    x = torch.randn(B, c, r, r, r, device="cuda")
    A = torch.rand(B, c, device="cuda") + 0.5; Bs = torch.randn(B, c, device="cuda") * 0.5
    items = B * (r ** 3 // 256)
    measure(f"plain {c}->{c} r{r}", lambda: conv3d_k3(x, conv1.weight, conv1.bias, split=True), items)
    measure(f"conv1 form (stats) dense {c} r{r}", lambda: fo.conv3d_fused(x, conv1, None, True, None), items)
    measure(f"conv2 form (pro+stats) dense {c} r{r}", lambda: fo.conv3d_fused(x, conv2, (A, Bs), True, None), items)
    coords = torch.randn(B, 3, n_pts, device="cuda"); feat = torch.randn(B, c, n_pts, device="cuda")
    out, _, _, cnt = bk.voxelize_points_forward(feat, coords, r, True, 0.0)
    grid = out.view(B, c, r, r, r)
    o1, o2 = fo.conv3d_occupancy(cnt, r, c, B)
    y1, _ = fo.conv3d_fused(grid, conv1, None, True, None)
    measure(f"conv1 sparse gauss {c} r{r}", lambda: fo.conv3d_fused(grid, conv1, None, True, fo.conv3d_occupancy(cnt, r, c, B)[0]), items)
    measure(f"conv2 delta gauss {c} r{r}", lambda: fo.conv3d_fused(y1, conv2, (A, Bs), True, fo.conv3d_occupancy(cnt, r, c, B)[1], prev_conv=conv1), items)
