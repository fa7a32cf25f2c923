#!/usr/bin/env bash
O=gpurun_out/r3c29; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
timeout 900 python -m pytest tests/test_fold_se_gpu.py tests/test_chain_gpu.py tests/test_models_gpu.py tests/test_b32_replay_gpu.py -x -q 2>&1 | tail -8
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dense-check > $O/bench_20.json
