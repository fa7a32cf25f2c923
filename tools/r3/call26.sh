#!/usr/bin/env bash
O=gpurun_out/r3c26; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
timeout 600 python -m pytest tests/test_conv_split_gpu.py tests/test_models_gpu.py tests/test_b32_replay_gpu.py -x -q 2>&1 | tail -6
for v in base cur base cur; do
  if [ $v = cur ]; then unset LION_HIP_SO; else export LION_HIP_SO=$PWD/tools/exp/variants/liblion_$v.so; fi
  timeout 200 python tools/conv_split_bench.py 2>&1 | grep "r 8" | sed "s/^/$v /"
done
unset LION_HIP_SO
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dense-check > $O/bench_cur.json
LION_HIP_SO=$PWD/tools/exp/variants/liblion_base.so timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dense-check > $O/bench_base.json
