#!/usr/bin/env bash
O=gpurun_out/r3c27; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
R=$PWD
for ev in 0 700; do
( cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr_$ev -o s -- python $R/tools/profile_step.py --only local --steps 3 --evolve $ev > /dev/null 2>&1 )
python tools/kstats.py $O/tr_$ev 40 --between-markers --per 3 > $O/kstats_evolve$ev.txt 2>&1
find $O/tr_$ev -name "*.csv" -size +20M -delete
done
for k in 100 300; do
timeout 300 python bench.py --gpus 1 --steps $k --warmup 5 --no-cpu-baseline --no-dense-check --repeats 1 > $O/bench_$k.json
done
