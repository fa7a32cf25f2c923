mkdir -p gpurun_out/r2b
./tools/exp/skinny_probe > gpurun_out/r2b/skinny_probe.txt 2>&1
bash tools/prof_traffic.sh devox_64_2048_32 devox -- python tools/one_devox.py 64 2048 32 > gpurun_out/r2b/devox_traffic.log 2>&1
bash tools/prof_traffic.sh global_prior skinny -- python tools/one_global_prior.py > gpurun_out/r2b/global_traffic.log 2>&1
bash tools/prof_traffic.sh vox_64_2048_32 vox_fused -- python tools/one_vox.py 64 2048 32 > gpurun_out/r2b/vox_traffic.log 2>&1
cat gpurun_out/r2b/skinny_probe.txt
