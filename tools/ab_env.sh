#!/usr/bin/env bash
# A/B of one environment switch on ONE box: usage tools/ab_env.sh VAR "v0 v1 ..." [extra bench args]; prints per value the
# driver-command step time (median of 3 x 20 steps), the real 1000-step chain and (with AB_BATCHES="32 4") other batches.
VAR=$1; VALS=$2; shift 2
for B in ${AB_BATCHES:-32}; do
  for rep in 1 2; do
    for v in $VALS; do
      env $VAR=$v python bench.py --gpus 1 --steps 20 --warmup 5 --batch $B --no-cpu-baseline --no-dense-check --forced-steps ${AB_FORCED:-0} --small-batches "" --detail-file gpurun_out/ab_detail.json "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v B=$B rep=$rep: ms_per_step %.3f  full_chain %.3f ms  forced %s' % (d['ms_per_step'], d.get('ms_per_step_full_chain') or 0, d.get('ms_per_step_forced_clouds')))"
    done
  done
done
