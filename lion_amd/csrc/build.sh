#!/usr/bin/env bash
# Builds liblion_hip.so for gfx950 (cross-compiles without a GPU).  -ffp-contract=off: the parity
# contract is "one IEEE rounding per written operation", same as the oracle.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -I../../include -Wall -Wno-unused-function"
OBJS=()
PIDS=()
for f in voxelize devoxelize ball_query grouping sampling interpolate chamfer emd diffusion conv3d conv3d_split conv3d_wgrad pointwise pwconv pwconv_split skinny attention "$@"; do
  [ -f "$f.hip" ] || continue
  if [ ! -f "$f.o" ] || [ "$f.hip" -nt "$f.o" ] || [ common.h -nt "$f.o" ] || [ split_ops.h -nt "$f.o" ] || [ ../../include/lion_hip.h -nt "$f.o" ]; then
    rm -f "$f.o"   # a failed compile must not leave the previous object behind to be linked silently
    $HIPCC $FLAGS -c "$f.hip" -o "$f.o" &
    PIDS+=($!)
  fi
  OBJS+=("$f.o")
done
for p in "${PIDS[@]}"; do wait "$p"; done   # set -e: any failed compile aborts the build
$HIPCC --offload-arch=gfx950 -shared -fPIC -o liblion_hip.so "${OBJS[@]}"
echo "built $(pwd)/liblion_hip.so"
