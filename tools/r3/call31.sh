#!/usr/bin/env bash
O=gpurun_out/r3c31; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
timeout 900 python -m pytest tests/test_scatter_csr_gpu.py tests/test_hip_parity_gpu.py tests/test_full_size_gpu.py -x -q 2>&1 | tail -12
timeout 200 python tools/kbench.py --only bwd 2>&1 | grep -v amdgpu
timeout 600 python bench.py --mode train_vae --no-cpu-baseline > $O/train_vae.json 2> $O/err.txt
timeout 600 python bench.py --mode train_prior --no-cpu-baseline > $O/train_prior.json 2>> $O/err.txt
