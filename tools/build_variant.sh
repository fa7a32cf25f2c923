#!/usr/bin/env bash
# A/B builds of liblion_hip.so: tools/exp/variants/liblion_NAME.so = the current library with some objects replaced.
#   tools/build_variant.sh NAME  file[@gitrev | =alternative/source.hip][:extra hipcc flags]  ...
# e.g.  tools/build_variant.sh pre_blind conv3d_split@026d1c7        (that file as of an older commit)
#       tools/build_variant.sh noslp 'sampling:-fno-slp-vectorize'    (the working-tree file with extra flags)
# Select at run time with LION_HIP_SO=tools/exp/variants/liblion_NAME.so (lion_amd/_lib.py).  The built .so files are
# git-ignored but travel to the GPU box with the gpurun snapshot.  Run lion_amd/csrc/build.sh first (the unchanged
# objects are taken from there).
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CSRC="$ROOT/lion_amd/csrc"
NAME="$1"; shift
OUT="$ROOT/tools/exp/variants"; mkdir -p "$OUT"
TMP="$(mktemp -d)"; trap 'rm -rf "$TMP"' EXIT
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fno-vectorize -munsafe-fp-atomics -I$ROOT/include -I$CSRC -Wall -Wno-unused-function"
declare -A REPL
PIDS=()
for spec in "$@"; do
  extra=""; if [[ "$spec" == *:* ]]; then extra="${spec#*:}"; spec="${spec%%:*}"; fi
  rev=""; if [[ "$spec" == *@* ]]; then rev="${spec#*@}"; spec="${spec%%@*}"; fi
  alt=""; if [[ "$spec" == *=* ]]; then alt="${spec#*=}"; spec="${spec%%=*}"; fi
  src="$TMP/$spec.hip"
  if [ -n "$rev" ]; then git -C "$ROOT" show "$rev:lion_amd/csrc/$spec.hip" > "$src"
  elif [ -n "$alt" ]; then cp "$alt" "$src"
  else cp "$CSRC/$spec.hip" "$src"; fi
  # shellcheck disable=SC2086
  $HIPCC $FLAGS $extra -c "$src" -o "$TMP/$spec.o" &
  PIDS+=($!)
  REPL[$spec]=1
done
for p in "${PIDS[@]}"; do wait "$p"; done
OBJS=()
for o in "$CSRC"/*.o; do
  b="$(basename "$o" .o)"
  if [ -n "${REPL[$b]:-}" ]; then OBJS+=("$TMP/$b.o"); else OBJS+=("$o"); fi
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/liblion_$NAME.so" "${OBJS[@]}"
echo "built $OUT/liblion_$NAME.so"
