#!/usr/bin/env bash
O=gpurun_out/r3c21; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
timeout 900 python -m pytest tests/test_training_gpu.py tests/test_models_gpu.py -x -q 2>&1 | tail -6
for p in 0 1; do
LION_TRAIN_PWCONV=$p timeout 600 python bench.py --mode train_prior --no-cpu-baseline > $O/train_prior_pw$p.json 2> $O/err.txt
LION_TRAIN_PWCONV=$p timeout 600 python bench.py --mode train_vae --no-cpu-baseline > $O/train_vae_pw$p.json 2> $O/err.txt
done
R=$PWD
( cd /tmp; export TMPDIR=/tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prior_trace -o prior -- python $R/bench.py --mode train_prior --steps 6 --warmup 2 --no-cpu-baseline > $R/$O/train_prior_traced.json 2> /dev/null )
python tools/kstats.py $O/prior_trace 40 > $O/prior_kernel_stats.txt 2>&1
find $O/prior_trace -name "*.csv" -size +20M -delete
