mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r2a/pytest.log
python bench.py --steps 100 --warmup 3 > gpurun_out/r2a/bench100.json 2> gpurun_out/r2a/bench100.err
LION_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 5 --warmup 1 --batch 4 --no-cpu-baseline --no-dense-check > gpurun_out/r2a/bench_spawn2.json 2> gpurun_out/r2a/bench_spawn2.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2a/prof -o step -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-dense-check > $GRAFT_REPO_ROOT/gpurun_out/r2a/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2a/prof.err
ls -R $GRAFT_REPO_ROOT/gpurun_out/r2a/prof | head
