#!/usr/bin/env bash
O=gpurun_out/r3c10; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
LION_HIP_SO=$PWD/tools/exp/variants/liblion_quad.so timeout 600 python -m pytest tests/test_conv_split_gpu.py -x -q 2>&1 | tail -5
for v in cur quad prio_taps prio_stage; do
  if [ $v = cur ]; then unset LION_HIP_SO; else export LION_HIP_SO=$PWD/tools/exp/variants/liblion_$v.so; fi
  timeout 200 python tools/conv_split_bench.py > $O/conv_split_bench_$v.txt 2>&1
  timeout 200 python tools/sparse_conv_bench.py > $O/sparse_conv_bench_$v.txt 2>&1
done
