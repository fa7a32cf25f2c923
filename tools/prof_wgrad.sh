# where the weight-gradient kernel's cycles go (64->64 @ 32^3, B=32): PMC counters in their own passes (no tracing)
# usage: tools/prof_wgrad.sh [CIN COUT R]
CIN=${1:-64}; COUT=${2:-64}; RR=${3:-32}
R=$PWD; O=$R/gpurun_out/wgradprof; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $O/pmc$i --output-format csv -- python $R/tools/one_wgrad.py $CIN $COUT $RR > $O/pmc$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for n in sorted(glob.glob("gpurun_out/wgradprof/pmc?")):
    fs = glob.glob(n + "/**/*counter_collection.csv", recursive=True)
    if not fs: print(n, "no output"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "wgrad_split" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print("%-28s launches %d mean %.4g" % (k, len(v), sum(v) / len(v)))
PY
