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

# The first ~36 searches of the run are the library's geometry trials (host.hip: GeoTuner) and the very first call
# is cold: the steady state is the LAST 100 dispatches of every kernel, from the kernel trace of the same run.
print("== steady state: the last 100 dispatches of each kernel (same run, kernel trace) ==")
for f in find("*kernel_trace.csv"):
    if os.sep + "trace" + os.sep not in f and not f.endswith(os.path.join("trace", "t_kernel_trace.csv")):
        continue
    per = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            per[row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]].append(
                (int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    for name, v in sorted(per.items(), key=lambda kv: -sum(d for _, d in kv[1])):
        v.sort()
        last = [d for _, d in v[-100:]]
        print(f"{name:60s} n={len(last)} avg_ns={sum(last) / len(last):.0f} min_ns={min(last)} max_ns={max(last)}")

print("== PMC (average per dispatch, kernels scan_kernel / filter_kernel / list_kernel) ==")
for f in find("*counter_collection.csv"):
    acc = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            kn = row.get("Kernel_Name", "")
            short = next((x for x in ("scan_kernel", "filter_kernel", "filter_dna_kernel", "list_kernel") if x in kn), None)
            if short is None:
                continue
            acc[(short, row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (kn, k), v in sorted(acc.items()):
        print(f"{kn}: {k} avg={sum(v) / len(v):.6g} n={len(v)}")


# HBM traffic per launch for bench.py's roofline.traffic (profiles/hbm_traffic.json).
# gfx950: FETCH_SIZE is in KB and reports 1/2 of the bytes of a wide coalesced streaming read
# (MI355X_MICROARCH.md, HBM section) -> read bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE * 1024.
import json

kern = defaultdict(dict)
for f in find("*counter_collection.csv"):
    acc = defaultdict(list)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            kn = row.get("Kernel_Name", "")
            short = next((x for x in ("scan_kernel", "filter_kernel", "filter_dna_kernel", "list_kernel",
                                      "trace_kernel", "build_chunks_kernel", "rank_count_kernel") if x in kn), None)
            if short:
                acc[(short, row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (kn, c), v in acc.items():
        kern[kn][c] = sum(v) / len(v)
if kern:
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/prof.sh",
           "correction": "read = FETCH_SIZE[KB]*1024*2 (gfx950 half-count of wide coalesced reads), write = WRITE_SIZE[KB]*1024",
           "text_bytes_per_gpu": int(os.environ.get("SASSY_PROF_TEXT_BYTES", "3000000000")),
           "kernels": {}}
    for kn, c in kern.items():
        rd = c.get("FETCH_SIZE", 0.0) * 1024 * 2
        wr = c.get("WRITE_SIZE", 0.0) * 1024
        out["kernels"][kn] = {"fetch_size_kb": c.get("FETCH_SIZE"), "write_size_kb": c.get("WRITE_SIZE"),
                              "hbm_bytes_per_launch": int(rd + wr)}
    with open(os.path.join(root, "hbm_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("== hbm_traffic.json ==")
    print(json.dumps(out, indent=1))
