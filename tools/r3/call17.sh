#!/usr/bin/env bash
O=gpurun_out/r3c17; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
R=$PWD
( cd /tmp; export TMPDIR=/tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/vae_trace -o vae -- python $R/bench.py --mode train_vae --steps 6 --warmup 2 --no-cpu-baseline > $R/$O/train_vae_traced.json 2> /dev/null )
python tools/kstats.py $O/vae_trace 60 > $O/vae_kernel_stats.txt 2>&1
find $O/vae_trace -name "*.csv" -size +20M -delete
