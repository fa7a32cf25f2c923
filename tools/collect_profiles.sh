# Round profile set (run through gpurun from the repo root; round 6: compact line + detail file, step census, training tables): everything the roofline numbers of DESIGN.md / bench.py are
# checked against.  Output: gpurun_out/profiles/<tag>_*; copy what should be judged into profiles/ and stamp it with the
# commit (tools/stamp_profiles.py -- the GPU box has no .git).
# The driver's command runs UNTRACED first (<tag>_bench_steps20_line.json): rocprofv3 changes how graphs replay (round 2's
# "9.22 ms" line had been taken under the tracer; untraced the same build gave 11.5), so the traced run of the same command is
# stored as <tag>_bench_steps20_TRACED_line.json and only serves the per-kernel tables.
# SHORT=1 (second argument "short"): only what a change of the sampling path's kernels moves -- the two bench lines, the traced step,
# operator / conv benches, traffic and MFMA-busy passes (training lines, probes and the 1x1 / wgrad benches keep their last set).
TAG=${1:-r06}
SHORT=0; [ "${2:-}" = short ] && SHORT=1
R=$PWD; O=$R/gpurun_out/profiles; mkdir -p $O
# the driver's command, untraced: the compact line on stdout + the full record (detail file); wall time of the command beside it
T0=$(python -c "import time; print(time.time())")
python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file $O/${TAG}_bench_steps20_detail.json > $O/${TAG}_bench_steps20_line.json 2> $O/${TAG}_bench_default.err
python -c "import time, sys; print('python bench.py --gpus 1 --steps 20 --warmup 5: %.1f s wall (process start to exit)' % (time.time() - float(sys.argv[1])))" $T0 > $O/${TAG}_bench_steps20_wall.txt
python bench.py --forced-steps 0 --small-batches "" --detail-file $O/${TAG}_bench_default_1000steps_detail.json > $O/${TAG}_bench_default_1000steps.json 2>> $O/${TAG}_bench_default.err
[ $SHORT = 1 ] || python bench.py --mode demo --detail-file $O/${TAG}_bench_detail.json > $O/${TAG}_bench_demo.json 2>> $O/${TAG}_bench_default.err
[ $SHORT = 1 ] || bash tools/train_profile.sh $TAG > $O/${TAG}_train_profile.log 2>&1
[ $SHORT = 1 ] || cp gpurun_out/train_prof/${TAG}_* $O/ 2>/dev/null
# kernel trace of the same command (rocprofv3 changes how graphs replay: its line is stored as *_TRACED_* and only serves the
# per-kernel tables) -> per-kernel stats, timeline and the launch census of one replayed step (B = 32, and B = 4 for strong scaling)
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/step_trace -o step -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-dense-check --no-full-chain --forced-steps 0 --small-batches "" --detail-file /tmp/_traced_detail.json > $O/${TAG}_bench_steps20_TRACED_line.json 2> /dev/null )
python tools/kstats.py $O/step_trace 70 > $O/${TAG}_bench_steps20_kernel_stats.txt 2>&1
python tools/trace_gaps.py $O/step_trace begin_step_kernel --top 45 --last 9 > $O/${TAG}_bench_steps20_timeline.txt 2>&1
python tools/step_census.py $O/step_trace --json $O/${TAG}_step_census_B32.json > $O/${TAG}_step_census_B32.txt 2>&1
# the global prior's 20 replayed steps of the same call sit in front of the local prior's 20
python tools/step_census.py $O/step_trace --slice -39 -21 --json $O/${TAG}_step_census_global_prior_B32.json > $O/${TAG}_step_census_global_prior_B32.txt 2>&1
cp $O/step_trace/step_kernel_stats.csv $O/${TAG}_bench_steps20_kernel_stats.csv 2>/dev/null
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --output-format csv -d $O/step_trace4 -o step -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --batch 4 --repeats 1 --no-cpu-baseline --no-dense-check --no-full-chain --forced-steps 0 --small-batches "" --detail-file /tmp/_traced_detail.json > /dev/null 2>&1 )
python tools/step_census.py $O/step_trace4 --json $O/${TAG}_step_census_B4.json > $O/${TAG}_step_census_B4.txt 2>&1
rm -rf $O/step_trace4
python tools/conv_split_bench.py > $O/${TAG}_conv_split_bench.txt 2>/dev/null
python tools/sparse_conv_bench.py > $O/${TAG}_sparse_conv_bench.txt 2>/dev/null
python tools/kbench.py > $O/${TAG}_kbench.txt 2>/dev/null
[ $SHORT = 1 ] || python tools/victims_beside_conv.py --replays 40 > $O/${TAG}_victims_beside_conv.txt 2>/dev/null
[ $SHORT = 1 ] || python tools/pw_bench.py > $O/${TAG}_pw_bench.txt 2>/dev/null
[ $SHORT = 1 ] || python tools/wgrad_bench.py > $O/${TAG}_wgrad_bench.txt 2>/dev/null
[ $SHORT = 1 ] || python tools/determinism_probe.py 32 5 > $O/${TAG}_determinism_probe.json 2>/dev/null
# HBM bytes per launch (FETCH_SIZE / WRITE_SIZE passes) of the kernels the bench line's rooflines are about -- the in-step
# forms: bench.py reads profiles/r*_{conv_instep,vox_scatter_64_2048_32,devox_affine_64_2048_32}_traffic.json into roofline.traffic
bash tools/prof_traffic.sh conv_instep conv3d_split_kernel -- python tools/one_conv_instep.py > /dev/null 2>&1
bash tools/prof_traffic.sh vox_scatter_64_2048_32 vox_scatter -- python tools/one_vox.py 64 2048 32 scatter > /dev/null 2>&1
[ $SHORT = 1 ] || bash tools/prof_traffic.sh vox_64_2048_32 vox_fused -- python tools/one_vox.py 64 2048 32 > /dev/null 2>&1
bash tools/prof_traffic.sh devox_affine_64_2048_32 devox_ring -- python tools/one_devox.py 64 2048 32 planned > /dev/null 2>&1
[ $SHORT = 1 ] || bash tools/prof_traffic.sh global_prior skinny -- python tools/one_global_prior.py > /dev/null 2>&1
for n in conv_instep vox_scatter_64_2048_32 vox_64_2048_32 devox_affine_64_2048_32 global_prior; do [ -f gpurun_out/traffic/$n.json ] && cp gpurun_out/traffic/$n.json $O/${TAG}_${n}_traffic.json; done
# MFMA-busy of the dominant conv on both kernels (own PMC passes, no tracing)
( cd /tmp; export TMPDIR=/tmp
  KS="split fp32 instep"; [ $SHORT = 1 ] && KS="split instep"
  for k in $KS; do
    S=1; [ $k = fp32 ] && S=0
    DRV="$R/tools/one_conv.py 64 64 32"; [ $k = instep ] && DRV="$R/tools/one_conv_instep.py"
    LION_CONV_SPLIT=$S timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --output-format csv -d $O/pmc_$k -- python $DRV > /dev/null 2>&1
    LION_CONV_SPLIT=$S timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$k -- python $DRV > /dev/null 2>&1
  done )
python - "$TAG" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]; O = "gpurun_out/profiles"
out = {}
for k in ("split", "fp32", "instep"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/pmc_{k}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv3d" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"] and "wmax" not in r["Kernel_Name"] and "wscale" not in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = []
    for f in glob.glob(f"{O}/trace_{k}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv3d_split_kernel" in r["Kernel_Name"] or "conv3d_k3_kernel" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    e = {c: sum(v) / len(v) for c, v in acc.items()}
    if dur: e["avg_us"] = sum(dur) / len(dur)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
        e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] * 128.0)  # GRBM_GUI_ACTIVE is summed over the 8 XCDs; x 32 CUs x 4 SIMDs each
    out[k] = e
json.dump({"kernel": "Conv3d 3x3x3 64->64 @32^3, B=32 (tools/one_conv.py)", "note": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE [sum over 8 XCDs] x 128 SIMDs per XCD); the counter pass runs at ~2.0 GHz (DVFS)", **out},
          open(f"{O}/{tag}_conv_mfma_util.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/step_trace $O/pmc_* $O/trace_*
ls -la $O
