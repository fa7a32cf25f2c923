R=$PWD; O=$R/gpurun_out/census${CENSUS_TAG:-}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for B in ${CENSUS_BATCHES:-32 4}; do
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_B$B -o step -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --batch $B --repeats 1 --no-cpu-baseline --no-dense-check --no-full-chain --forced-steps 0 --small-batches "" > $O/line_B$B.json 2> /dev/null
  python $R/tools/step_census.py $O/trace_B$B --json $O/census_B$B.json > $O/census_B$B.txt 2>&1
  python $R/tools/trace_gaps.py $O/trace_B$B begin_step_kernel --top 60 --last 9 > $O/timeline_B$B.txt 2>&1
  rm -rf $O/trace_B$B
done
head -3 $O/census_B32.txt $O/census_B4.txt
