#!/usr/bin/env bash
mkdir -p gpurun_out/profiles
exec > gpurun_out/profiles/collect_log.txt 2>&1
set -x
bash tools/collect_profiles.sh r03
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
