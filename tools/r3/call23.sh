#!/usr/bin/env bash
O=gpurun_out/r3c23; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
for c in 8 12 16 24 32 64; do
LION_VOX_CHCAP=$c LION_HIP_SO=$PWD/tools/exp/variants/liblion_sweep.so timeout 100 python tools/kbench.py --only vox 2>&1 | grep "(64, 2048, 32)" | sed "s/^/chcap $c /"
done
for v in cur sweep cur sweep; do
  if [ $v = cur ]; then unset LION_HIP_SO; else export LION_HIP_SO=$PWD/tools/exp/variants/liblion_$v.so; fi
  timeout 200 python tools/conv_split_bench.py 2>&1 | grep "r32\|r16" | sed "s/^/$v /"
  timeout 200 python tools/sparse_conv_bench.py 2>&1 | grep "C=" | sed "s/^/$v /"
done
