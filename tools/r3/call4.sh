#!/usr/bin/env bash
# round 3, GPU call 4: -fno-slp-vectorize for the whole library (A/B on kernel times and the step), eager multi-stream
# variants (what real overlap is worth), FPS sharing its CUs (no whole-LDS request) with the no-SLP build
O=gpurun_out/r3c4; mkdir -p $O
V=$PWD/tools/exp/variants
exec > $O/log.txt 2>&1
set -x
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dense-check"
S="LION_GEOMETRY_PREFETCH=0 LION_OVERLAP_POINT_BRANCH=0"
env $S timeout 300 $B > $O/bench_ss_head.json
env $S LION_HIP_SO=$V/liblion_all_noslp.so timeout 300 $B > $O/bench_ss_all_noslp.json
env $S LION_HIP_SO=$V/liblion_side_noslp.so timeout 300 $B > $O/bench_ss_side_noslp.json
LION_HIP_SO=$V/liblion_all_noslp.so timeout 300 python tools/conv_split_bench.py > $O/conv_split_bench_all_noslp.txt
timeout 300 python tools/conv_split_bench.py > $O/conv_split_bench_head.txt
LION_HIP_SO=$V/liblion_all_noslp.so timeout 300 python tools/pw_bench.py > $O/pw_bench_all_noslp.txt
timeout 300 python tools/pw_bench.py > $O/pw_bench_head.txt
LION_HIP_SO=$V/liblion_all_noslp.so timeout 300 python tools/kbench.py > $O/kbench_all_noslp.txt
timeout 300 python tools/kbench.py > $O/kbench_head.txt
# eager (no graph): real streams
LION_GEOMETRY_PREFETCH=1 LION_OVERLAP_POINT_BRANCH=0 timeout 300 $B --no-graph > $O/bench_eager_geo1_pt0.json
LION_GEOMETRY_PREFETCH=0 LION_OVERLAP_POINT_BRANCH=0 timeout 300 $B --no-graph > $O/bench_eager_geo0_pt0.json
LION_GEOMETRY_PREFETCH=1 LION_OVERLAP_POINT_BRANCH=1 timeout 300 $B --no-graph > $O/bench_eager_geo1_pt1.json
LION_FPS_SHARE_CU=1 LION_HIP_SO=$V/liblion_side_noslp.so LION_GEOMETRY_PREFETCH=1 LION_OVERLAP_POINT_BRANCH=0 timeout 300 $B --no-graph > $O/bench_eager_geo1_pt0_share.json
LION_FPS_SHARE_CU=1 LION_HIP_SO=$V/liblion_side_noslp.so LION_GEOMETRY_PREFETCH=1 LION_OVERLAP_POINT_BRANCH=1 timeout 300 $B > $O/bench_graph_geo1_pt1_share.json
# all victims beside the convolution with the no-SLP library, FPS sharing CUs
LION_FPS_SHARE_CU=1 LION_HIP_SO=$V/liblion_all_noslp.so timeout 600 python tools/victims_beside_conv.py --replays 100 > $O/victims_all_noslp_share.txt
date
