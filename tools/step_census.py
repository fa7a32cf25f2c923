"""Kernel census of ONE replayed sampling step from a rocprofv3 --kernel-trace CSV: launches per step and time per step of every
kernel between consecutive launches of MARKER (begin_step_kernel: first kernel of a chain step), over the last K intervals (the
local prior's chain runs last).  ATen kernels (at::native / at::cuda) are listed separately: bench.py reads the JSON this writes
(profiles/r*_step_census*.json) into config.launches_per_step / config.aten_kernels_in_step.
usage: step_census.py DIR [--last K | --first K] [--json OUT] [--top N] [--marker NAME]"""
import csv, glob, json, re, sys
from collections import defaultdict

d = sys.argv[1]
arg = lambda k, dv: (sys.argv[sys.argv.index(k) + 1] if k in sys.argv else dv)
last, top, marker = int(arg("--last", 9)), int(arg("--top", 80)), arg("--marker", "begin_step_kernel")
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:70]


rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f))]
ts = sorted(s for s, e, k in rows if marker in k)
first = int(arg("--first", 0))      # --first K: the first K intervals instead (the global prior's chain runs first)
ts = ts[:first + 1] if first else ts[-(last + 1):]
if "--slice" in sys.argv:            # --slice A B: markers[A:B] of the whole list (python slice; e.g. -39 -21 = the global prior's
    i_ = sys.argv.index("--slice")   # replays of a 20-step timed call: the local prior's 20 steps come after them)
    ts = sorted(s for s, e, k in rows if marker in k)[int(sys.argv[i_ + 1]):int(sys.argv[i_ + 2])]
lo, hi, n = ts[0], ts[-1], len(ts) - 1
cnt, tim = defaultdict(int), defaultdict(float)
for s, e, k in rows:
    if lo <= s < hi:
        cnt[k] += 1
        tim[k] += (e - s) / 1e3
is_aten = lambda k: k.startswith("at::") or "at::native" in k or "at::cuda" in k
total = sum(cnt.values()) / n
aten = {k: cnt[k] / n for k in cnt if is_aten(k)}
print(f"{n} steps, span {(hi - lo) / 1e3 / n:.0f} us per step; {total:.1f} launches per step, {sum(aten.values()):.1f} of them ATen; "
      f"sum of kernel durations {sum(tim.values()) / n:.0f} us per step")
print(f"{'kernel':70s} {'n/step':>7s} {'us/step':>9s} {'us/launch':>10s}")
for k in sorted(cnt, key=lambda k: -tim[k])[:top]:
    print(f"{k:70s} {cnt[k] / n:7.1f} {tim[k] / n:9.1f} {tim[k] / cnt[k]:10.1f}")
out = arg("--json", "")
if out:
    json.dump({"steps": n, "us_per_step_span": (hi - lo) / 1e3 / n, "launches_per_step": total,
               "aten_kernels_per_step": sum(aten.values()), "aten_names": aten,
               "kernels": {k: {"per_step": cnt[k] / n, "us_per_step": tim[k] / n} for k in sorted(cnt, key=lambda k: -tim[k])}},
              open(out, "w"), indent=1)
