#!/usr/bin/env bash
# tools/exp/liblion_timing.so = liblion_hip.so with s_memtime phase counters compiled into an instrumented COPY of
# csrc/conv3d_split.hip (the product file only carries "// @phase N" comments at the phase boundaries) and the -DPWS_TIMING
# build of csrc/pwconv_split.hip.  Run the readers with
#   LION_HIP_SO=$PWD/tools/exp/liblion_timing.so python tools/conv_phase_times.py | tools/pw_phase_times.py
# The counters stay in registers until a workgroup ends: a memory operation per mark would sit in front of every vmcnt wait
# of the kernel and be measured instead of it.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/lion_amd/csrc"
bash build.sh > /dev/null
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fno-vectorize -munsafe-fp-atomics -I../../include -I."
python3 - <<'PY'
import re
s = open("conv3d_split.hip").read()
hdr = '''
__device__ unsigned long long g_split_phase[8];
__device__ unsigned long long g_split_clk[2]; // sum over workgroups of (s_memtime ticks, 100 MHz wall ticks) of the workgroup's lifetime
__device__ unsigned long long g_split_hist[8]; // durations of one tap group (phase 6: TG taps between two group barriers)
#define PH_MARK(k) PH_MARKH(k, false)
#define PH_MARKH(k, H) do { const unsigned long long n_ = __builtin_readcyclecounter(); const unsigned d_ = (unsigned)(n_ - t_ph); ph_acc[k] += d_; \
  if (H) { const unsigned per_ = d_ / (unsigned)PH_GROUP_MFMAS; ph_hist[per_ < 36 ? 0 : per_ < 42 ? 1 : per_ < 50 ? 2 : per_ < 58 ? 3 : per_ < 66 ? 4 : per_ < 76 ? 5 : per_ < 95 ? 6 : 7] += 1; } \
  t_ph = n_; } while (0)
'''
s = s.replace('namespace {\n', 'namespace {\n' + hdr, 1)
s = s.replace('// @phase-init', 'unsigned long long t_ph = __builtin_readcyclecounter(); unsigned ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned ph_hist[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long c_ph0 = t_ph, w_ph0 = wall_clock64();')
s = s.replace('// @phase-flush', 'if (tid == 0) { atomicAdd(&g_split_clk[0], __builtin_readcyclecounter() - c_ph0); atomicAdd(&g_split_clk[1], wall_clock64() - w_ph0); } if (tid == 0) for (int kk_ = 0; kk_ < 8; ++kk_) { atomicAdd(&g_split_phase[kk_], (unsigned long long)ph_acc[kk_]); atomicAdd(&g_split_hist[kk_], (unsigned long long)ph_hist[kk_]); }')
s = s.replace('        // @phase 6\n        asm volatile("s_waitcnt vmcnt(0)"', '        PH_MARKH(6, k >= 1);\n        asm volatile("s_waitcnt vmcnt(0)"', 1)
s = s.replace('template <int TD, int TH, int TW, int CB, int VB, bool PRO, bool STATS, int OCC>\n__global__', '#define PH_GROUP_MFMAS (TG * 3 * CB * VB)\ntemplate <int TD, int TH, int TW, int CB, int VB, bool PRO, bool STATS, int OCC>\n__global__', 1)
s = s.replace('template <bool PRO, bool STATS, int NW>\n__global__', '#undef PH_GROUP_MFMAS\n#define PH_GROUP_MFMAS 27\ntemplate <bool PRO, bool STATS, int NW>\n__global__', 1)
s = re.sub(r'// @phase (\d)', r'PH_MARK(\1);', s)
s = s.replace('// @phase-reader', '''int lion_debug_split_phases(unsigned long long *host8, int reset) {
  if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_split_phase), 64) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_split_phase), z, 64) != hipSuccess) return -1; }
  return 0;
}
int lion_debug_split_clk(unsigned long long *host2, int reset) {
  if (hipMemcpyFromSymbol(host2, HIP_SYMBOL(g_split_clk), 16) != hipSuccess) return -1;
  if (reset) { unsigned long long z[2] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_split_clk), z, 16) != hipSuccess) return -1; }
  return 0;
}
int lion_debug_split_hist(unsigned long long *host8, int reset) {
  if (hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_split_hist), 64) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_split_hist), z, 64) != hipSuccess) return -1; }
  return 0;
}''')
open("/tmp/lion_conv3d_split_timing.hip", "w").write(s)
PY
/opt/rocm/bin/hipcc $F -c /tmp/lion_conv3d_split_timing.hip -o /tmp/lion_cs_timing.o
/opt/rocm/bin/hipcc $F -DPWS_TIMING -c pwconv_split.hip -o /tmp/lion_pws_timing.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/liblion_timing.so \
  $(ls *.o | grep -v -e '^conv3d_split.o$' -e '^pwconv_split.o$') /tmp/lion_cs_timing.o /tmp/lion_pws_timing.o
echo "built tools/exp/liblion_timing.so"
