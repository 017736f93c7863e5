#!/bin/bash
# tools/pmc_cmd.sh <kernel name substring> <command ...> -- ON THE GPU BOX: instruction / wave / stall counters of one kernel of
# any command, averaged per dispatch (counters in their own runs, --kernel-trace only)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
KERN=$1; shift
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LEVEL_WAVES" \
            "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_IFETCH GRBM_GUI_ACTIVE" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT64 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  out=gpurun_out/pmc_cmd/$(echo $pass | cut -c1-12 | tr ' ' _)
  rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --pmc $pass -f csv -d $out -o p -- "$@" > /dev/null 2> $out/err.txt
  python - "$out" "$KERN" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
root, kern = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
if not acc: print("no counters:", open(os.path.join(root, "err.txt")).read()[-400:])
print(kern, {k: round(sum(v[-20:]) / len(v[-20:]), 1) for k, v in sorted(acc.items())})
PY
done
