#!/usr/bin/env bash
O=gpurun_out/r3c24; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20.json
timeout 200 python tools/kbench.py --only vox > $O/kbench_vox.txt 2>&1
