#!/usr/bin/env bash
# round 3, GPU call 3: (1) FPS: s_setprio / a single packed-fp32 expression; (2) why is the untraced multi-branch graph
# replay 2.4 ms per step slower than under rocprofv3: HIP runtime graph knobs
O=gpurun_out/r3c3; mkdir -p $O
V=$PWD/tools/exp/variants
exec > $O/log.txt 2>&1
set -x
for v in fps_slp_nosetprio fps_noslp_onepk fps_noslp_onepk_nosetprio; do
  LION_FPS_SHARE_CU=1 LION_HIP_SO=$V/liblion_$v.so timeout 300 python tools/victims_beside_conv.py --replays 40 fps > $O/victims_$v.txt
done
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dense-check"
timeout 300 $B > $O/bench_default.json
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 300 $B > $O/bench_pktcap0.json
DEBUG_HIP_FORCE_GRAPH_QUEUES=1 timeout 300 $B > $O/bench_gq1.json
DEBUG_HIP_FORCE_GRAPH_QUEUES=2 timeout 300 $B > $O/bench_gq2.json
DEBUG_HIP_FORCE_GRAPH_QUEUES=4 timeout 300 $B > $O/bench_gq4.json
GPU_STREAMOPS_CP_WAIT=1 timeout 300 $B > $O/bench_cpwait1.json
GPU_STREAMOPS_CP_WAIT=0 timeout 300 $B > $O/bench_cpwait0.json
GPU_MAX_HW_QUEUES=8 timeout 300 $B > $O/bench_hwq8.json
DEBUG_HIP_GRAPH_BATCH_SIZE=4096 timeout 300 $B > $O/bench_batch4096.json
DEBUG_HIP_DYNAMIC_QUEUES=0 timeout 300 $B > $O/bench_dynq0.json
timeout 300 $B > $O/bench_default_b.json
LION_GEOMETRY_PREFETCH=0 LION_OVERLAP_POINT_BRANCH=0 timeout 300 $B > $O/bench_single_stream.json
date
