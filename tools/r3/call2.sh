#!/usr/bin/env bash
# round 3, GPU call 2: (1) which property of the SLP build of fps_reg_kernel makes it fail beside the split convolution:
# packed fp32 arithmetic or the 194-register footprint; (2) untraced step time with / without the side streams
O=gpurun_out/r3c2; mkdir -p $O
V=$PWD/tools/exp/variants
exec > $O/log.txt 2>&1
set -x
for v in fps_noslp_fp200 fps_noslp_fp256 fps_slp_le128 fps_slp_fp256; do
  LION_FPS_SHARE_CU=1 LION_HIP_SO=$V/liblion_$v.so timeout 300 python tools/victims_beside_conv.py --replays 40 fps > $O/victims_$v.txt
done
LION_FPS_SHARE_CU=1 timeout 300 python tools/victims_beside_conv.py --replays 20 fps > $O/victims_stock.txt
for cfg in 1:1 0:1 1:0 0:0; do
  g=${cfg%%:*}; p=${cfg##*:}
  LION_GEOMETRY_PREFETCH=$g LION_OVERLAP_POINT_BRANCH=$p timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dense-check > $O/bench_geo${g}_pt${p}.json
done
LION_GEOMETRY_PREFETCH=1 LION_OVERLAP_POINT_BRANCH=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dense-check --no-graph > $O/bench_eager.json
timeout 300 python bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --no-dense-check > $O/bench_100.json
date
