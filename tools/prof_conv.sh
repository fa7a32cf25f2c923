# MFMA utilisation of the dominant conv (64->64 @ 32^3, B=32) from PMC counters (own pass, no tracing)
R=$PWD; O=$R/gpurun_out/convprof; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/tools/one_conv.py 64 64 32 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d $O/pmc1 --output-format csv -- python $R/tools/one_conv.py 64 64 32 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/pmc2 --output-format csv -- python $R/tools/one_conv.py 64 64 32 > /dev/null 2>&1
cd $R; python tools/kstats.py $O/trace 3 --per 5 | grep -i conv3d
python - <<'PY'
import csv, glob, collections
for n in ("pmc1", "pmc2"):
    fs = glob.glob(f"gpurun_out/convprof/{n}/**/*counter_collection.csv", recursive=True)
    if not fs: print(n, "no output"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "conv3d_k3_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(n, k, "launches", len(v), "mean", sum(v) / len(v))
PY
