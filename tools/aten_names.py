"""Full template names of the ATen kernels in a rocprofv3 --kernel-trace CSV directory (kstats.py cuts names at 64 characters,
which hides the functor): count, average and total microseconds per name.  --tail F keeps only the last fraction F of the launches (a bench trace ends with the timed replays: --tail 0.1 of a
"--steps 5 --warmup 3" trace is about one replayed step, without the eager warm-up / capture passes).
usage: aten_names.py DIR [top_n] [--per N] [--tail F]"""
import csv, glob, re, sys
from collections import defaultdict
d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 30
per = float(sys.argv[sys.argv.index("--per") + 1]) if "--per" in sys.argv else 1.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
agg = defaultdict(lambda: [0, 0.0])
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
if "--tail" in sys.argv:
    rows = rows[-int(len(rows) * float(sys.argv[sys.argv.index("--tail") + 1])):]
print("%d launches considered, %.0f us of kernels" % (len(rows), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e3))
for r in rows:
    n = r["Kernel_Name"]
    if "at::native" not in n and "rocclr" not in n:
        continue
    n = re.sub(r"^void ", "", n).replace("at::native::", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)
    agg[n][0] += 1; agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print("ATen / runtime kernels: %.0f launches, %.0f us (per %g)" % (sum(v[0] for v in agg.values()) / per, tot / per, per))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%7.1f %8.1f %10.1f  %s" % (n / per, t / n, t / per, k[:230]))
