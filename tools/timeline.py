"""Lone-search timeline from a rocprofv3 --kernel-trace CSV: per kernel the average duration over the last N searches,
the idle gaps between consecutive kernels of one search, and the turnaround between two searches (last kernel end ->
next search's first kernel start = host wake-up + result handling + enqueue).
Usage: python tools/timeline.py <dir with *kernel_trace.csv> [first-kernel-substring]"""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
first_key = sys.argv[2] if len(sys.argv) > 2 else "filter_dna_kernel"
files = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("sassy_hip::", "")[:48]))
rows.sort()
# searches = runs of kernels that start with first_key
starts = [i for i, r in enumerate(rows) if first_key in r[2]]
searches = [rows[a:b] for a, b in zip(starts, starts[1:] + [len(rows)])]
searches = searches[-60:-5] if len(searches) > 70 else searches
dur = defaultdict(list); gap = defaultdict(list); turn = []; span = []
for i, s in enumerate(searches):
    for j, (a, b, n) in enumerate(s):
        dur[(j, n)].append(b - a)
        if j:
            gap[(j, n)].append(a - s[j - 1][1])
    span.append(s[-1][1] - s[0][0])
    if i + 1 < len(searches):
        turn.append(searches[i + 1][0][0] - s[-1][1])
avg = lambda v: sum(v) / max(1, len(v)) / 1e3
print(f"{len(searches)} searches; first kernel start -> last kernel end: {avg(span):.1f} us; turnaround to the next search: {avg(turn):.1f} us")
for (j, n), v in sorted(dur.items()):
    g = gap.get((j, n))
    print(f"  {j} {n:48s} n={len(v):3d} dur {avg(v):8.1f} us   gap before {avg(g) if g else 0:6.1f} us")
