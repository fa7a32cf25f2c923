"""Print a compact per-kernel table from a rocprofv3 --kernel-trace --stats CSV directory."""
import csv, glob, sys, re
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)[:60]
    print(f"{name:60s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} {float(r['MinNs'])/1e3:9.1f} {float(r['MaxNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}")
