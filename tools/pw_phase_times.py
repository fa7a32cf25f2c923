"""Per-chunk timeline (s_memtime) of wave 0 of workgroup 0 of the split-operand 1x1 convolution: wait, barrier, prefetch issue,
cut, MFMAs, then the epilogue (needs tools/exp/liblion_timing.so, see tools/build_timing_lib.sh)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lion_amd import _lib, fused_ops as fo
lib = _lib.load()
lib.lion_debug_pws_times.restype = ctypes.c_int
lib.lion_debug_pws_times.argtypes = [ctypes.c_void_p]
for cin, cout, L in [(128, 128, 16), (192, 128, 2048)]:
    conv = torch.nn.Conv1d(cin, cout, 1).cuda(); x = torch.randn(32, cin, L, device="cuda")
    with torch.no_grad():
        for _ in range(3): fo.pwconv_fused(x, conv, None, split=True)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 128)()
    lib.lion_debug_pws_times(buf)
    t0 = buf[0]
    n = (cin + 15) // 16
    print(f"{cin}->{cout} L={L}: prologue issued {buf[1]-t0}")
    for q in range(n + 4):
        if buf[2 + 4*q] == 0 or 5 + 4*q >= 100: break
        print(f"  chunk {q}: wait_done +{buf[2+4*q]-t0}  barrier +{buf[3+4*q]-buf[2+4*q]}  issue +{buf[4+4*q]-buf[3+4*q]}  cut +{(buf[64+q]-buf[4+4*q]) if q < n else 0} mfma +{(buf[5+4*q]-buf[64+q]) if q < n else 0}")
    print(f"  loop end {buf[100]-t0}  stores +{buf[102]-buf[100]}  stats regs +{buf[103]-buf[102]}  stats out +{buf[101]-buf[103]}")
