mkdir -p gpurun_out/r2e
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r2e/pytest.log
./tools/exp/sparse_rows_probe > gpurun_out/r2e/sparse_rows.txt 2>&1
python bench.py --steps 100 --warmup 3 --no-cpu-baseline > gpurun_out/r2e/bench100.json 2> gpurun_out/r2e/bench100.err
cat gpurun_out/r2e/pytest.log gpurun_out/r2e/sparse_rows.txt; head -c 1500 gpurun_out/r2e/bench100.json
