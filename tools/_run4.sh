mkdir -p gpurun_out/r2d
python -m pytest tests/test_hip_parity_gpu.py tests/test_full_size_gpu.py -m gpu -q -x -k "devox" 2>&1 | tail -8 > gpurun_out/r2d/pytest_devox.log
bash tools/prof_traffic.sh devox_64_2048_32 devox -- python tools/one_devox.py 64 2048 32 > gpurun_out/r2d/devox_traffic.log 2>&1
python tools/kbench.py --only devox > gpurun_out/r2d/kbench_devox.txt 2>&1
python tools/exp/conv3d_split_check.py > gpurun_out/r2d/split_check.txt 2>&1
python tools/exp/conv3d_split_check.py --model > gpurun_out/r2d/split_check_model.txt 2>&1
cat gpurun_out/r2d/pytest_devox.log gpurun_out/r2d/kbench_devox.txt gpurun_out/r2d/split_check.txt gpurun_out/r2d/split_check_model.txt
