#!/usr/bin/env bash
O=gpurun_out/r3c8; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
timeout 600 python -m pytest tests/test_conv_split_gpu.py tests/test_b32_replay_gpu.py tests/test_concurrency_gpu.py -x -q 2>&1 | tail -5
for v in base cur; do
  if [ $v = cur ]; then unset LION_HIP_SO; else export LION_HIP_SO=$PWD/tools/exp/variants/liblion_$v.so; fi
  timeout 200 python tools/conv_split_bench.py > $O/conv_split_bench_$v.txt 2>&1
  timeout 200 python tools/sparse_conv_bench.py > $O/sparse_conv_bench_$v.txt 2>&1
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$v.json
done
