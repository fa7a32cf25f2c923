#!/usr/bin/env bash
O=gpurun_out/r3c14; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
LION_HIP_SO=$PWD/tools/exp/liblion_timing.so timeout 300 python tools/conv_phase_times.py > $O/conv_phase_times.txt 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_20.json
timeout 600 python -m pytest tests/test_bench_gpu.py -x -q 2>&1 | tail -3
