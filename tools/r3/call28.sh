#!/usr/bin/env bash
O=gpurun_out/r3c28; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
timeout 900 python -m pytest tests/test_hip_parity_gpu.py tests/test_models_gpu.py tests/test_training_gpu.py tests/test_bench_gpu.py -x -q 2>&1 | tail -6
for v in base cur; do
  if [ $v = cur ]; then unset LION_HIP_SO; else export LION_HIP_SO=$PWD/tools/exp/variants/liblion_$v.so; fi
  timeout 200 python tools/wgrad_bench.py > $O/wgrad_bench_$v.txt 2>&1
done
unset LION_HIP_SO
timeout 600 python bench.py --mode train_vae --no-cpu-baseline > $O/train_vae.json 2> $O/err.txt
timeout 600 python bench.py --mode train_prior --no-cpu-baseline > $O/train_prior.json 2>> $O/err.txt
