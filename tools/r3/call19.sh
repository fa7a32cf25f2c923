#!/usr/bin/env bash
O=gpurun_out/r3c19; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
timeout 600 python -m pytest tests/test_conv_split_gpu.py tests/test_pwconv_split_gpu.py tests/test_training_gpu.py tests/test_train_ops_gpu.py -x -q 2>&1 | tail -5
LION_BENCH_ADAM_FUSED=0 timeout 600 python bench.py --mode train_vae --no-cpu-baseline > $O/train_vae_adam0.json 2> $O/train_vae_adam0.err
timeout 600 python bench.py --mode train_vae --no-cpu-baseline > $O/train_vae.json 2> $O/train_vae.err
timeout 600 python bench.py --mode train_prior --no-cpu-baseline > $O/train_prior.json 2> $O/train_prior.err
