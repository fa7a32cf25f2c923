#!/usr/bin/env bash
# tools/exp/liblion_timing.so = liblion_hip.so with the s_memtime phase counters of the split-operand kernels compiled in
# (-DSPLIT_EXP_TIMING: csrc/conv3d_split.hip, -DPWS_TIMING: csrc/pwconv_split.hip).  Run the readers with
#   LION_HIP_SO=$PWD/tools/exp/liblion_timing.so python tools/conv_phase_times.py | tools/pw_phase_times.py
# The counters stay in registers until a workgroup ends: a memory operation per mark would sit in front of every vmcnt wait
# of the kernel and be measured instead of it.
set -euo pipefail
cd "$(dirname "$0")/../lion_amd/csrc"
bash build.sh > /dev/null
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I../../include"
/opt/rocm/bin/hipcc $F -DSPLIT_EXP_TIMING -c conv3d_split.hip -o /tmp/lion_cs_timing.o
/opt/rocm/bin/hipcc $F -DPWS_TIMING -c pwconv_split.hip -o /tmp/lion_pws_timing.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/liblion_timing.so \
  $(ls *.o | grep -v -e '^conv3d_split.o$' -e '^pwconv_split.o$') /tmp/lion_cs_timing.o /tmp/lion_pws_timing.o
echo "built tools/exp/liblion_timing.so"
