#!/bin/bash
# tools/prof_rc.sh -- ON THE GPU BOX: rocprofv3 kernel stats of tools/probe_rc.py (forward-only and both-strand
# searches of the 3 GB text) -> gpurun_out/prc/summary.txt
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prc
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -f csv -d $OUT -o t -- python tools/probe_rc.py > $OUT/out.json 2>/dev/null
python - "$OUT" > $OUT/summary.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
d = defaultdict(list)
for f in set(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("# tools/probe_rc.py under rocprofv3 --kernel-trace --stats: 3 GB, |pattern| = 32, k = 3; Dna and Iupac")
print("# searchers, forward strand only and both strands (6 calls each).  Both strands = ONE filter pass")
print("# (filter_count_kernel marks the candidate blocks of both strands) + two chunk-list / DP / rank / trace")
print("# chains on two lanes; no reverse kernel, no second pass over the text.")
print(open(os.path.join(root, "out.json")).read().strip())
print("kernel | launches | avg us")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k} | {len(v)} | {sum(v) / len(v) / 1e3:.1f}")
PY
cat $OUT/summary.txt
