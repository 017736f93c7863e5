#!/bin/bash
# tools/pmc_ab.sh [PROBE_* settings] -- ON THE GPU BOX: instruction / wave counters of the dominant kernel of tools/probe_fused.py,
# previous round's library against this tree's (counters in their own runs, --kernel-trace only)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for lib in libsassy_hip_r3.so libsassy_hip.so; do
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LEVEL_WAVES"; do
    out=gpurun_out/pmc_ab/$lib/$(echo $pass | cut -c1-12 | tr ' ' _)
    rm -rf $out; mkdir -p $out
    env "$@" SASSY_HIP_LIBRARY=$PWD/sassy_amd/lib/$lib rocprofv3 --kernel-trace --pmc $pass -f csv -d $out -o p -- python tools/probe_fused.py > /dev/null 2> $out/err.txt
    python tools/pmc_avg.py "$out" "$lib"
  done
done
