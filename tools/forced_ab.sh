#!/usr/bin/env bash
# round-5 review, next 2: which round-5 optimisations survive on the clouds a TRAINED model visits?  The product sampler with the
# local prior's latent forced to sqrt(abar_t) S + sqrt(1 - abar_t) z before every step (bench.py ForcedClouds), 1000 steps,
# same box: default / LION_CONV_SKIP_UNREAD=0 / --no-sparse / the scatter without adoption (tools/build_variant.sh noadopt).
run() {  # label, env..., -- extra args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-dense-check --small-batches "" --detail-file gpurun_out/ab_detail.json "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('%-34s forced clouds %.3f ms/step = %.2f shapes/s (empty tiles conv1 %.2f conv2 %.2f) | own chain %.3f ms/step = %.2f shapes/s (empty %.2f / %.2f) | 20-step %.3f' % ('$label', d['ms_per_step_forced_clouds'], d['value_forced_clouds'], c['forced_clouds_conv1_empty_tile_frac'], c['forced_clouds_conv2_empty_tile_frac'], d['ms_per_step_full_chain'], d['value_full_chain_1000'], c['chain_conv1_empty_tile_frac'], c['chain_conv2_empty_tile_frac'], d['ms_per_step']))"
}
for rep in 1 2; do
run "default" X=1 --
run "LION_CONV_SKIP_UNREAD=0" LION_CONV_SKIP_UNREAD=0 --
run "LION_CONV_SKIP_UNREAD_LEVEL2=0" LION_CONV_SKIP_UNREAD_LEVEL2=0 --
run "scatter without adoption" LION_HIP_SO=tools/exp/variants/liblion_noadopt.so --
run "--no-sparse (every tile computed)" X=1 -- --no-sparse
done
