#!/usr/bin/env bash
O=gpurun_out/r3c30; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
timeout 900 python -m pytest tests/test_hip_parity_gpu.py tests/test_full_size_gpu.py tests/test_training_gpu.py -x -q 2>&1 | tail -5
for v in base cur; do
  if [ $v = cur ]; then unset LION_HIP_SO; else export LION_HIP_SO=$PWD/tools/exp/variants/liblion_$v.so; fi
  timeout 200 python tools/kbench.py --only bwd 2>&1 | grep -v amdgpu | sed "s/^/$v /"
done
