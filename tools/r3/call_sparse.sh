O=gpurun_out/pc_sparse.txt; : > $O
for cfg in "64 32 2048 4 1 conv1" "64 32 2048 32 1 conv1" "64 32 2048 32 1 conv2" "128 16 1024 32 1 conv1" "128 16 1024 32 0 conv1" "64 32 2048 32 0 conv2"; do
  echo "== $cfg" >> $O
  timeout 30 python tools/r3/pc_sparse_case.py $cfg >> $O 2>&1; echo "rc=$?" >> $O
done
