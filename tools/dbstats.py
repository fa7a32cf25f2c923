"""Per-kernel table from a rocprofv3 rocpd database (*_results.db; what `rocprofv3 --kernel-trace` writes when no
--output-format csv is given).  usage: dbstats.py FILE.db [top_n] [--per N] [--last-frac F]
--last-frac F keeps only the dispatches in the last fraction F of the trace's time span (drops warm-up / capture)."""
import re, sqlite3, sys
from collections import defaultdict

db = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 40
per = float(sys.argv[sys.argv.index("--per") + 1]) if "--per" in sys.argv else 1.0
frac = float(sys.argv[sys.argv.index("--last-frac") + 1]) if "--last-frac" in sys.argv else 1.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, start, end from kernels").fetchall()
lo, hi = min(r[1] for r in rows), max(r[2] for r in rows)
cut = hi - (hi - lo) * frac
rows = [r for r in rows if r[1] >= cut]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:70]
agg = defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    k = short(n); agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
span = (max(r[2] for r in rows) - min(r[1] for r in rows)) / 1e3
print(f"kernels: {len(rows)}  sum of kernel time: {tot/per:.0f} us  wall span: {span/per:.0f} us  (per {per:g})")
print(f"{'kernel':70s} {'calls':>8s} {'avg_us':>9s} {'tot_us':>10s} {'pct':>6s}")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k:70s} {n/per:8.1f} {t/n:9.1f} {t/per:10.1f} {100*t/tot:6.2f}")
