#!/usr/bin/env bash
# Kernel-level evidence for the training configurations (BASELINE.json configs[2..4]; round-5 review, missing 4): the three
# --mode train_* lines untraced, then a rocprofv3 kernel trace of the same command per mode -> top-45 kernel table
# (tools/kstats.py) under gpurun_out/train_prof/<tag>_train_<mode>_kernel_stats.txt.  usage: tools/train_profile.sh TAG [modes]
TAG=${1:-r06}; shift || true
MODES=${*:-train_vae train_prior train_prior_clip}
R=$PWD; O=$R/gpurun_out/train_prof; mkdir -p $O
for m in $MODES; do
  # the VAE step takes no vendor-library path at all: LION_STRICT=1 turns any into an error (the priors' global denoiser keeps its
  # 2048-wide [32-column] layers on the rocBLAS matrix product: profiles/r06_train_pwconv_ab.txt)
  STRICT=0; [ $m = train_vae ] && STRICT=1
  LION_STRICT=$STRICT python bench.py --mode $m --steps 5 --warmup 3 --detail-file $O/${TAG}_bench_detail.json > $O/${TAG}_bench_${m}.json 2> $O/${TAG}_bench_${m}.err
  mv $O/${TAG}_bench_detail_${m}.json $O/${TAG}_bench_${m}_detail.json 2>/dev/null
  ( cd /tmp; export TMPDIR=/tmp
    rocprofv3 --kernel-trace --output-format csv -d $O/trace_$m -o t -- python $R/bench.py --mode $m --steps 5 --warmup 3 --no-cpu-baseline --detail-file /tmp/_d.json > /dev/null 2>&1 )
  # the last 5 replays of the step graph: between the first and last launch of the step's first kernel is fragile across modes;
  # the whole trace (capture warm-ups included) / per-launch averages are what the table reports
  python tools/kstats.py $O/trace_$m 45 > $O/${TAG}_train_${m}_kernel_stats.txt 2>&1
  rm -rf $O/trace_$m
  python - <<PY
import json
d = json.loads(open("$O/${TAG}_bench_${m}.json").read().strip().splitlines()[-1])
print("$m: %.1f ms/step, %.0f samples/s, fallbacks %s, strict %s" % (d["ms_per_step"], d["value"], d["config"].get("vendor_library_fallbacks_total"), d["config"].get("strict")))
PY
done
