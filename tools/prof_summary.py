"""Summarise rocprofv3 CSV output of tools/prof.sh: per-kernel time stats and per-dispatch PMC
averages for the scan kernel.  Usage: python tools/prof_summary.py gpurun_out/prof_<tag>"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("*kernel_stats.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Name", "")[:60]
            print(f"{name:60s} calls={row.get('Calls')} avg_ns={row.get('AverageNs')} "
                  f"min_ns={row.get('MinNs')} max_ns={row.get('MaxNs')} pct={row.get('Percentage')}")

print("== PMC (average per dispatch, kernels scan_kernel / filter_kernel / list_kernel) ==")
for f in find("*counter_collection.csv"):
    acc = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            kn = row.get("Kernel_Name", "")
            short = next((x for x in ("scan_kernel", "filter_kernel", "list_kernel") if x in kn), None)
            if short is None:
                continue
            acc[(short, row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (kn, k), v in sorted(acc.items()):
        print(f"{kn}: {k} avg={sum(v) / len(v):.6g} n={len(v)}")
