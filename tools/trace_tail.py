"""The last N kernel dispatches of a rocprofv3 --kernel-trace run, in time order, with durations and the idle gap in
front of each, then per-kernel totals.  Usage: python tools/trace_tail.py <dir> [N]"""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = []
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                         r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("sassy_hip::", "").split("(")[0][:56]))
rows.sort()
tail = rows[-N:]
prev_end = None
for a, b, nm in tail:
    gap = (a - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"  {nm:56s} dur {(b - a) / 1e3:9.1f} us   gap before {gap:8.1f} us")
    prev_end = max(prev_end or 0, b)
tot = defaultdict(lambda: [0, 0])
for a, b, nm in rows:
    tot[nm][0] += 1
    tot[nm][1] += b - a
print("-- totals over the run")
for nm, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {nm:56s} calls {c:5d}  total {t / 1e6:9.3f} ms  avg {t / c / 1e3:9.1f} us")
