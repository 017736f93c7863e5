"""Average per dispatch of every counter of a rocprofv3 --pmc run for the streaming kernels (last 20 dispatches)."""
import csv, glob, sys, os
from collections import defaultdict
root, lib = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"]
        if "filter_dna_kernel" in kn or "filter_count" in kn or "scan_kernel" in kn:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(lib, {k: round(sum(v[-20:]) / len(v[-20:]), 1) for k, v in sorted(acc.items())})
