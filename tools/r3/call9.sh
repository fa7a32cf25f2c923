#!/usr/bin/env bash
O=gpurun_out/r3c9; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
LION_HIP_SO=$PWD/tools/exp/liblion_timing.so timeout 300 python tools/conv_phase_times.py > $O/conv_phase_times.txt 2>&1
