#!/usr/bin/env bash
O=gpurun_out/r3c20; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
timeout 600 python -m pytest tests/test_train_ops_gpu.py -x -q 2>&1 | tail -15
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_golden_gpu.py tests/test_models_gpu.py -x -q 2>&1 | tail -6
timeout 600 python bench.py --mode train_vae --no-cpu-baseline > $O/train_vae.json 2> $O/train_vae.err
timeout 600 python bench.py --mode train_prior --no-cpu-baseline > $O/train_prior.json 2> $O/train_prior.err
timeout 600 python bench.py --mode train_prior_clip --no-cpu-baseline > $O/train_prior_clip.json 2> $O/train_prior_clip.err
