#!/usr/bin/env bash
O=gpurun_out/r3c32; mkdir -p $O
exec > $O/log.txt 2>&1
set -x
timeout 900 python -m pytest tests/test_scatter_csr_gpu.py -x -q 2>&1 | tail -5
R=$PWD
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/tr -o s -- python $R/tools/kbench.py --only bwd > /dev/null 2>&1 )
python tools/kstats.py $O/tr 14 > $O/kstats.txt 2>&1
rm -rf $O/tr
timeout 200 python tools/kbench.py --only bwd 2>&1 | grep -v amdgpu
