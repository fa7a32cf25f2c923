#!/usr/bin/env bash
O=gpurun_out/r3c6; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
timeout 200 python tools/conv_split_bench.py > $O/conv_split_bench.txt 2>&1
timeout 200 python tools/sparse_conv_bench.py > $O/sparse_conv_bench.txt 2>&1
timeout 200 python tools/pw_bench.py > $O/pw_bench.txt 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20.json
