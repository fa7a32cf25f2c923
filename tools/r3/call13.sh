#!/usr/bin/env bash
O=gpurun_out/r3c13; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
for d in noprio samestream; do
LION_SPLIT_DEBUG=$d timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dense-check --repeats 1 > $O/bench_$d.json
done
LION_GEOMETRY_SPLIT_GRAPH=0 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dense-check --repeats 1 > $O/bench_split0.json
