"""Where a wave of the split-operand convolution spends its cycles: per-phase s_memtime totals of wave 0 of every workgroup
(needs tools/exp/liblion_timing.so, see tools/build_timing_lib.sh).  r >= 16: conv3d_split_kernel; r = 8: the pipelined kernel
(phases 1 / 5 / 6 / 4 = weight wait / group barrier / taps / staging)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd import _lib
from lion_amd.conv_ops import conv3d_k3
lib = _lib.load()
lib.lion_debug_split_phases.restype = ctypes.c_int
lib.lion_debug_split_phases.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["item prologue", "barrier A (chunk start)", "loads + wait + activate + max", "barrier B (max)", "cut + LDS write",
         "group barrier (weights)", "taps of a group", "epilogue"]
for cin, cout, r in [(128, 128, 8), (64, 64, 32)]:
    conv = torch.nn.Conv3d(cin, cout, 3, padding=1).cuda(); x = torch.randn(32, cin, r, r, r, device="cuda")
    with torch.no_grad():
        for _ in range(3): conv3d_k3(x, conv.weight, conv.bias, split=True)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        lib.lion_debug_split_phases(buf, 1)
        n = 5
        for _ in range(n): conv3d_k3(x, conv.weight, conv.bias, split=True)
        torch.cuda.synchronize()
        lib.lion_debug_split_phases(buf, 1)
    tot = sum(buf)
    print(f"{cin}->{cout} r{r}: total wave-0 cycles per launch {tot / n:.3e}")
    for k in range(8): print(f"   {names[k]:34s} {100.0 * buf[k] / tot:6.2f} %")
