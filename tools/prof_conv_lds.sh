# LDS / issue counters of the dominant forward convolution (64->64 @ 32^3, B = 32, split kernel): PMC passes only (no tracing)
# usage: tools/prof_conv_lds.sh [instep]
DRV="tools/one_conv.py 64 64 32"; [ "${1:-}" = instep ] && DRV="tools/one_conv_instep.py"
R=$PWD; O=$R/gpurun_out/convlds; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $O/pmc$i --output-format csv -- python $R/$DRV > $O/pmc$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for n in sorted(glob.glob("gpurun_out/convlds/pmc?")):
    fs = glob.glob(n + "/**/*counter_collection.csv", recursive=True)
    if not fs: print(n, "no output"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "conv3d_split_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print("%-28s launches %d mean %.4g" % (k, len(v), sum(v) / len(v)))
PY
