mkdir -p gpurun_out/r2c
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r2c/pytest.log
./tools/exp/skinny_probe 2>&1 | head -3 > gpurun_out/r2c/skinny_probe.txt
bash tools/prof_traffic.sh devox_64_2048_32 devox -- python tools/one_devox.py 64 2048 32 > gpurun_out/r2c/devox_traffic.log 2>&1
bash tools/prof_traffic.sh global_prior skinny -- python tools/one_global_prior.py > gpurun_out/r2c/global_traffic.log 2>&1
python bench.py --steps 50 --warmup 3 --no-cpu-baseline --no-dense-check > gpurun_out/r2c/bench50.json 2> gpurun_out/r2c/bench50.err
cat gpurun_out/r2c/pytest.log gpurun_out/r2c/skinny_probe.txt
