#!/usr/bin/env bash
O=gpurun_out/r3c15; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
LION_HIP_SO=$PWD/tools/exp/variants/liblion_vox2.so timeout 600 python -m pytest tests/test_hip_parity_gpu.py tests/test_full_size_gpu.py -x -q -k "vox" 2>&1 | tail -4
for v in base cur vox2; do
  if [ $v = cur ]; then unset LION_HIP_SO; else export LION_HIP_SO=$PWD/tools/exp/variants/liblion_$v.so; fi
  timeout 200 python tools/kbench.py --only vox > $O/kbench_vox_$v.txt 2>&1
  if [ $v != vox2 ]; then
  timeout 200 python tools/conv_split_bench.py > $O/conv_split_bench_$v.txt 2>&1
  timeout 200 python tools/sparse_conv_bench.py > $O/sparse_conv_bench_$v.txt 2>&1
  fi
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dense-check > $O/bench_$v.json
done
unset LION_HIP_SO
timeout 600 python -m pytest tests/test_conv_split_gpu.py -x -q 2>&1 | tail -3
