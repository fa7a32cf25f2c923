# usage: tools/prof_traffic.sh NAME KERNEL_SUBSTR -- python tools/one_xxx.py args
# Three separate rocprofv3 passes of the same command (kernel trace; --pmc FETCH_SIZE; --pmc WRITE_SIZE -- the two
# counters do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") and a JSON summary for the kernels whose
# name contains KERNEL_SUBSTR: average duration, raw counters (KiB units) and HBM bytes per launch with the gfx950
# correction (FETCH_SIZE x 2 for wide coalesced reads).  Output: gpurun_out/traffic/NAME.json (+ the CSVs).
NAME=$1; SUB=$2; shift 3
R=$PWD; O=$R/gpurun_out/traffic/$NAME; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp   # (a fresh directory: the summary below reads the first CSV it finds)
( cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- "$@" > /dev/null 2>&1 )
( cd $R && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- "$@" > /dev/null 2>&1 )
( cd $R && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- "$@" > /dev/null 2>&1 )
cd $R; python - "$NAME" "$SUB" <<'PY'
import csv, glob, json, re, sys, collections
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:70]
name, sub = sys.argv[1], sys.argv[2]
O = f"gpurun_out/traffic/{name}"
out = {"name": name, "kernel_filter": sub, "command": "see tools/prof_traffic.sh", "kernels": {}}
f = glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)
dur = collections.defaultdict(list)
if f:
    for r in csv.DictReader(open(f[0])):
        if sub in r["Kernel_Name"]:
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
ctr = {}
for n in ("fetch", "write"):
    f = glob.glob(O + f"/{n}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            if sub in r["Kernel_Name"]:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    ctr[n] = acc
for k, v in dur.items():
    fe, wr = ctr["fetch"].get(k, []), ctr["write"].get(k, [])
    e = {"launches": len(v), "avg_us": sum(v) / len(v), "min_us": min(v)}
    if fe: e["FETCH_SIZE_KiB_avg"] = sum(fe) / len(fe)
    if wr: e["WRITE_SIZE_KiB_avg"] = sum(wr) / len(wr)
    if fe and wr:
        e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE_KiB_avg"] + e["WRITE_SIZE_KiB_avg"]) * 1024.0
        e["note"] = "FETCH_SIZE doubled (gfx950 tallies 128-B requests as 64 B on wide coalesced reads); WRITE_SIZE as reported"
    out["kernels"][k] = e
json.dump(out, open(O + ".json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
