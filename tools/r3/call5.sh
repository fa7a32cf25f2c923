#!/usr/bin/env bash
O=gpurun_out/r3c5; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
timeout 180 python tools/r3/pc_bringup.py
echo "bringup rc=$?"
timeout 300 python tools/conv_split_bench.py
LION_HIP_SO=$PWD/tools/exp/variants/liblion_all_noslp.so timeout 300 python tools/conv_split_bench.py
timeout 300 python tools/sparse_conv_bench.py
timeout 600 python -m pytest tests/test_conv_split_gpu.py -x -q 2>&1 | tail -15
