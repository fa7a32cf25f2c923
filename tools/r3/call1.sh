#!/usr/bin/env bash
# round 3, GPU call 1: diagnostics at HEAD -- (1) the new B=32 replay tests, (2) the driver's bench command against the
# library with conv3d_split.hip as of 026d1c7 / d99d308 (the two blind commits of round 2), (3) FPS beside the split
# convolution with and without SLP-packed fp32 arithmetic, the generic victim harness, the MFMA victim matrix,
# (4) kernel trace of the 20-step chain.  Everything lands in gpurun_out/r3c1/.
O=gpurun_out/r3c1; mkdir -p $O
V=tools/exp/variants
exec > $O/log.txt 2>&1
set -x
date
timeout 900 python -m pytest tests/test_b32_replay_gpu.py -x -q 2>&1 | tail -15
date
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_head_w5.json
for tag in head:w5 head:w2 pre_blind:w5 dma_builtin:w5 head:w5b pre_blind:w5b; do
  lib=${tag%%:*}; w=${tag##*:}; W=5; [ $w = w2 ] && W=2
  so=""; [ $lib != head ] && so=$PWD/$V/liblion_$lib.so
  LION_HIP_SO=$so timeout 300 python bench.py --gpus 1 --steps 20 --warmup $W --no-cpu-baseline > $O/bench_${lib}_${w}_nocpu.json
done
date
timeout 300 python tools/conv_split_bench.py > $O/conv_split_bench_head.txt
LION_HIP_SO=$PWD/$V/liblion_pre_blind.so timeout 300 python tools/conv_split_bench.py > $O/conv_split_bench_pre_blind.txt
timeout 300 python tools/sparse_conv_bench.py > $O/sparse_conv_bench_head.txt
LION_HIP_SO=$PWD/$V/liblion_pre_blind.so timeout 300 python tools/sparse_conv_bench.py > $O/sparse_conv_bench_pre_blind.txt
date
# FPS sharing its CUs with the convolution: stock build (SLP-packed fp32) vs -fno-slp-vectorize build of sampling.hip
LION_FPS_SHARE_CU=1 timeout 300 python tools/victims_beside_conv.py --replays 40 fps > $O/victims_fps_share_stock.txt
LION_FPS_SHARE_CU=1 LION_HIP_SO=$PWD/$V/liblion_noslp.so timeout 300 python tools/victims_beside_conv.py --replays 100 fps > $O/victims_fps_share_noslp.txt
LION_FPS_SHARE_CU=1 timeout 300 python tools/victims_beside_conv.py --replays 40 --B 2 fps > $O/victims_fps_share_stock_B2.txt
LION_FPS_SHARE_CU=1 LION_HIP_SO=$PWD/$V/liblion_noslp.so timeout 300 python tools/victims_beside_conv.py --replays 100 --B 2 fps > $O/victims_fps_share_noslp_B2.txt
timeout 600 python tools/victims_beside_conv.py --replays 100 > $O/victims_all_default.txt
date
timeout 120 ./tools/exp/mfma_matrix > $O/mfma_victim_matrix.txt
timeout 120 ./tools/exp/mfma_probe > $O/mfma_neighbour_probe.txt
date
R=$PWD
( cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/step_trace -o step -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense-check > $R/$O/bench_traced_line.json 2> /dev/null )
python tools/kstats.py $O/step_trace 80 > $O/step_kernel_stats.txt 2>&1
python tools/trace_gaps.py $O/step_trace begin_step_kernel --top 45 --last 9 > $O/step_timeline.txt 2>&1
cp $O/step_trace/*kernel_stats.csv $O/step_kernel_stats.csv 2>/dev/null
find $O/step_trace -name "*.csv" -size +20M -delete
date
