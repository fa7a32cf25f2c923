mkdir -p gpurun_out/r2f
for v in NO_GATHER NO_DMA CT_4 CT_16 CT_2; do
  echo "== $v" >> gpurun_out/r2f/devox_variants.txt
  LION_HIP_SO=$PWD/tools/exp/liblion_devox_$v.so python tools/kbench.py --only devox 2>&1 | grep "2048, 32" >> gpurun_out/r2f/devox_variants.txt
done
echo "== product" >> gpurun_out/r2f/devox_variants.txt
python tools/kbench.py --only devox 2>&1 | grep "2048, 32" >> gpurun_out/r2f/devox_variants.txt
cat gpurun_out/r2f/devox_variants.txt
