#!/bin/bash
# tools/prof.sh <tag> [bench args...] -- rocprofv3 kernel stats + PMC passes of bench.py on the GPU box.
# Counters are collected in their own runs (never together with sys/hip/hsa tracing).
# Outputs CSVs under gpurun_out/prof_<tag>/ and a compact summary gpurun_out/prof_<tag>/summary.txt
set -u
TAG=${1:-x}; shift || true
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
# (counter passes: no geometry trials, the byte and instruction counts do not depend on them; the timed stats run
# keeps the tuner on -- prof_summary.py reports the last 100 dispatches of every kernel as the steady state)
BENCH="env SASSY_HIP_TUNE=0 python bench.py --steps 3 --warmup 1 --tune-searches 0 --no-cpu-baseline $*"
BENCH_STATS="python bench.py --steps 100 --warmup 20 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- $BENCH_STATS > $OUT/bench_trace.json 2> $OUT/trace.err
pass() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -f csv -d $OUT/pmc_$name -o p -- $BENCH > /dev/null 2> $OUT/pmc_$name.err
}
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD
pass b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU
pass c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH SQ_LEVEL_WAVES
pass d FETCH_SIZE
pass e WRITE_SIZE
python tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
