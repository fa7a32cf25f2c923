# per-kernel time + HBM traffic (PMC, separate passes) of voxelize_points (C=64, N=2048, r=32, B=32)
R=$PWD; O=$R/gpurun_out/voxprof; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/tools/one_vox.py 64 2048 32 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/tools/one_vox.py 64 2048 32 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/tools/one_vox.py 64 2048 32 > /dev/null 2>&1
cd $R; python tools/kstats.py $O/trace 4 --per 10
python - <<'PY'
import csv, glob
for n in ("fetch", "write"):
    f = glob.glob(f"gpurun_out/voxprof/{n}/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "vox_fused" in r["Kernel_Name"]]
    print(n, "launches", len(v), "mean counter (KB units per guide)", sum(v) / len(v))
PY
