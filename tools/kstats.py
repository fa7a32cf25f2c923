"""Compact per-kernel table from a rocprofv3 --kernel-trace CSV directory.
usage: kstats.py DIR [top_n] [--between-markers [MARKER]] [--per N] [--by-grid]
--between-markers keeps only kernels launched between the first and last launch of MARKER (default
chamfer_fwd_kernel); --per N divides totals by N (e.g. steps) to print per-step microseconds; --by-grid keeps
launches of one kernel with different grids apart (the grid, in workgroups, is appended to the name)."""
import csv, glob, re, sys
from collections import defaultdict

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 40
marker = None
if "--between-markers" in sys.argv:
    i = sys.argv.index("--between-markers")
    marker = sys.argv[i + 1] if len(sys.argv) > i + 1 and not sys.argv[i + 1].startswith("--") else "chamfer_fwd_kernel"
per = float(sys.argv[sys.argv.index("--per") + 1]) if "--per" in sys.argv else 1.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:64]
if marker:
    ts = [int(r["Start_Timestamp"]) for r in rows if marker in r["Kernel_Name"]]
    lo, hi = min(ts), max(ts)
    rows = [r for r in rows if lo < int(r["Start_Timestamp"]) < hi]
agg = defaultdict(lambda: [0, 0.0])
for r in rows:
    k = short(r["Kernel_Name"])
    if "--by-grid" in sys.argv:
        g = [int(r.get("Grid_Size_" + a, 0) or 0) // max(int(r.get("Workgroup_Size_" + a, 1) or 1), 1) for a in "XYZ"]
        k = k[:44] + " [%d,%d,%d]" % tuple(g)
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[k][0] += 1; agg[k][1] += dur
tot = sum(v[1] for v in agg.values())
span = (max(int(r["End_Timestamp"]) for r in rows) - min(int(r["Start_Timestamp"]) for r in rows)) / 1e3 if rows else 0
print(f"kernels: {len(rows)}  sum of kernel time: {tot/per:.0f} us  wall span: {span/per:.0f} us  (per {per:g})")
print(f"{'kernel':64s} {'calls':>7s} {'avg_us':>9s} {'tot_us':>10s} {'pct':>6s}")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k:64s} {n/per:7.1f} {t/n:9.1f} {t/per:10.1f} {100*t/tot:6.2f}")
