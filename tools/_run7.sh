mkdir -p gpurun_out/r2g
python -m pytest tests -m gpu -q -x -k "devox or skinny or full_batch or priors" 2>&1 | tail -6 > gpurun_out/r2g/pytest.log
./tools/exp/skinny_probe 2>&1 | grep -v batch-major > gpurun_out/r2g/skinny_probe.txt
python tools/kbench.py --only devox > gpurun_out/r2g/kbench_devox.txt 2>&1
bash tools/prof_traffic.sh global_prior skinny -- python tools/one_global_prior.py > gpurun_out/r2g/global_traffic.log 2>&1
cat gpurun_out/r2g/pytest.log gpurun_out/r2g/skinny_probe.txt gpurun_out/r2g/kbench_devox.txt; grep -A4 '"skinny_gemm\|"skinny_fin' gpurun_out/traffic/global_prior.json | grep -E "kernel|avg_us|launches"
