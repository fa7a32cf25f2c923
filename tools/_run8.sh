mkdir -p gpurun_out/r2h
python -m pytest tests -m gpu -q -x -k "devox or full_batch" 2>&1 | tail -4 > gpurun_out/r2h/pytest.log
python tools/kbench.py --only devox > gpurun_out/r2h/kbench_devox.txt 2>&1
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2h/prof -o step -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-dense-check > $R/gpurun_out/r2h/prof_bench.json 2> $R/gpurun_out/r2h/prof.err
cd $R
cat gpurun_out/r2h/pytest.log gpurun_out/r2h/kbench_devox.txt
python tools/kstats.py gpurun_out/r2h/prof 45 > gpurun_out/r2h/kstats.txt 2>&1; cat gpurun_out/r2h/kstats.txt
