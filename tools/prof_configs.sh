#!/bin/bash
# tools/prof_configs.sh <tag> -- rocprofv3 kernel stats + VALU / HBM counters for BASELINE configs 3 and 4
# (tools/bench_configs.py; config 4 on $PATTERNS patterns, default 20: one chain per pattern; PATTERNS=10000 CONFIGS=4
# profiles the seeded search).  Counters in their own runs, never mixed with
# API tracing.  Output: gpurun_out/prof_<tag>/summary.txt
set -u
TAG=${1:-cfg}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python tools/bench_configs.py --configs ${CONFIGS:-3,4} --patterns ${PATTERNS:-20} --steps 3"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- $CMD > $OUT/bench.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_a -o p -- $CMD > /dev/null 2> $OUT/pmc_a.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_d -o p -- $CMD > /dev/null 2> $OUT/pmc_d.err
python - "$OUT" > $OUT/summary.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
dur = defaultdict(list)
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
pmc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        pmc[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel | launches | avg us | VALU wave-instr per launch | VALU lane-ops/s (x64 lanes) | HBM read GB/s (FETCH_SIZE*2048/avg)")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if "sassy_hip" not in k:
        continue
    avg = sum(v) / len(v)
    valu = pmc[k].get("SQ_INSTS_VALU")
    fetch = pmc[k].get("FETCH_SIZE")
    vi = sum(valu) / len(valu) if valu else float("nan")
    fb = sum(fetch) / len(fetch) * 2048 if fetch else float("nan")
    print(f"{k[:70]:70s} | {len(v):4d} | {avg / 1e3:9.1f} | {vi:12.4g} | {vi * 64 / (avg * 1e-9):10.3g} | {fb / avg:8.1f}")
PY
cat $OUT/summary.txt
