#!/usr/bin/env bash
O=gpurun_out/r3c11; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
for v in base cur; do
  if [ $v = cur ]; then unset LION_HIP_SO; else export LION_HIP_SO=$PWD/tools/exp/variants/liblion_$v.so; fi
  timeout 200 python tools/conv_split_bench.py > $O/conv_split_bench_$v.txt 2>&1
  timeout 200 python tools/sparse_conv_bench.py > $O/sparse_conv_bench_$v.txt 2>&1
  timeout 200 python tools/pw_bench.py > $O/pw_bench_$v.txt 2>&1
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$v.json
done
unset LION_HIP_SO
R=$PWD
( cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/step_trace -o step -- python $R/bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-dense-check > $R/$O/bench_traced_line.json 2> /dev/null )
python tools/kstats.py $O/step_trace 80 > $O/step_kernel_stats.txt 2>&1
python tools/trace_gaps.py $O/step_trace begin_step_kernel --top 60 --last 9 > $O/step_timeline.txt 2>&1
find $O/step_trace -name "*.csv" -size +20M -delete
