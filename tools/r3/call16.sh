#!/usr/bin/env bash
O=gpurun_out/r3c16; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
timeout 600 python -m pytest tests/test_train_ops_gpu.py -x -q 2>&1 | tail -15
timeout 600 python -m pytest tests/test_hip_parity_gpu.py tests/test_full_size_gpu.py -x -q -k "vox" 2>&1 | tail -3
for f in 0 1; do
LION_TRAIN_FUSE=$f timeout 600 python bench.py --mode train_vae --no-cpu-baseline > $O/train_vae_fuse$f.json 2> $O/train_vae_fuse$f.err
LION_TRAIN_FUSE=$f timeout 600 python bench.py --mode train_prior --no-cpu-baseline > $O/train_prior_fuse$f.json 2> $O/train_prior_fuse$f.err
done
