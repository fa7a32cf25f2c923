#!/usr/bin/env bash
O=gpurun_out/r3c18; mkdir -p $O
exec > $O/log.txt 2>&1
timeout 600 python tools/train_op_profile.py > $O/train_op_profile.txt 2>&1
