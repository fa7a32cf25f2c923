for m in train_vae train_prior; do
 for v in ${AB_COLS:-262144 65536 8192}; do
  LION_TRAIN_PWCONV_MIN_COLS=$v python bench.py --mode $m --steps 5 --warmup 3 --no-cpu-baseline --detail-file gpurun_out/ab_detail.json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$m LION_TRAIN_PWCONV_MIN_COLS=$v: %.1f ms/step, vendor-library notes %s' % (d['ms_per_step'], d['config'].get('vendor_library_fallbacks_total')))"
  python -c "
import json
d=json.load(open('gpurun_out/ab_detail_$m.json'))
for k,v in (d['config'].get('vendor_library_fallback_shapes') or {}).items(): print('     %4d  %s' % (v,k))"
 done
done
