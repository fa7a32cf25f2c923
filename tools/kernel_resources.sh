#!/usr/bin/env bash
# Per-kernel register / scratch / LDS / occupancy table of every HIP source of the library, from the compiler's own
# resource remarks (-Rpass-analysis=kernel-resource-usage).  No GPU needed.  usage: tools/kernel_resources.sh > profiles/<tag>_kernel_resources.txt
set -euo pipefail
cd "$(dirname "$0")/../lion_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fno-vectorize -munsafe-fp-atomics -I../../include"
printf "%-18s %-78s %5s %5s %8s %4s %8s\n" file kernel VGPR AGPR scratch occ LDS_static
for f in *.hip; do
  /opt/rocm/bin/hipcc $F -c "$f" -Rpass-analysis=kernel-resource-usage -o /tmp/_kr.o 2>&1 | python3 -c '
import re, subprocess, sys
name = sys.argv[1]
cur = None
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"n": m.group(1)}
        continue
    if cur is None: continue
    for key, pat in (("v", r"\bVGPRs: (\d+)"), ("a", r"AGPRs: (\d+)"), ("s", r"ScratchSize \[bytes/lane\]: (\d+)"), ("o", r"Occupancy \[waves/SIMD\]: (\d+)"), ("l", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m: cur[key] = m.group(1)
    if "l" in cur:
        try:
            dem = subprocess.run(["c++filt", cur["n"]], capture_output=True, text=True).stdout.strip()
        except Exception:
            dem = cur["n"]
        dem = re.sub(r"\(anonymous namespace\)::", "", dem); dem = re.sub(r"^void ", "", dem); dem = re.sub(r"\(.*", "", dem)
        print("%-18s %-78s %5s %5s %8s %4s %8s" % (name, dem[:78], cur.get("v", "?"), cur.get("a", "?"), cur.get("s", "?"), cur.get("o", "?"), cur["l"]))
        cur = None
' "$f"
done
rm -f /tmp/_kr.o
