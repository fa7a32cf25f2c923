"""Where the wall time of a replayed step goes: union of busy intervals of a rocprofv3 --kernel-trace CSV between the
first and last launch of MARKER, idle gaps (attributed to the kernel that ends the gap), and per-kernel EXCLUSIVE time
(the time a kernel is the only one running; concurrent time is shared equally).
usage: trace_gaps.py DIR [MARKER] [--top N] [--last K]   (--last K: only the last K marker intervals, i.e. the replayed
steps of the chain that ran last, without the python time between runs)"""
import csv, glob, re, sys
from collections import defaultdict
d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "begin_step_kernel"
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 30
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:60]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f))]
ts = [s for s, e, k in rows if marker in k]
ts.sort()
if "--last" in sys.argv: ts = ts[-(int(sys.argv[sys.argv.index("--last") + 1]) + 1):]
lo, hi = min(ts), max(ts)
rows = sorted(r for r in rows if lo <= r[0] < hi)
ev = []
for i, (s, e, k) in enumerate(rows): ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active, excl, gap_by, last, idle = set(), defaultdict(float), defaultdict(lambda: [0, 0.0]), lo, 0.0
for t, kind, i in ev:
    dt = t - last
    if dt > 0:
        if active:
            for j in active: excl[rows[j][2]] += dt / len(active)
        elif kind == 1:
            idle += dt; g = gap_by[rows[i][2]]; g[0] += 1; g[1] += dt
    last = t
    if kind == 1: active.add(i)
    else: active.discard(i)
span = hi - lo
n = len(ts) - 1
print(f"span {span/1e3:.0f} us over {n} marker intervals ({span/1e3/n:.0f} us each); busy {100*(span-idle)/span:.1f} %, idle {100*idle/span:.1f} % "
      f"in {sum(g[0] for g in gap_by.values())} gaps (avg {idle/1e3/max(1,sum(g[0] for g in gap_by.values())):.2f} us)")
print(f"{'kernel':60s} {'share_us/int':>12s} {'pct':>6s}   | gaps before it: {'n/int':>6s} {'us/int':>8s}")
for k, t in sorted(excl.items(), key=lambda kv: -kv[1])[:top]:
    g = gap_by.get(k, [0, 0.0])
    print(f"{k:60s} {t/1e3/n:12.1f} {100*t/span:6.2f}   | {g[0]/n:21.1f} {g[1]/1e3/n:8.1f}")
