#!/bin/bash
# tools/ab.sh [rounds] [PROBE_* settings ...] -- ON THE GPU BOX: a lone search (tools/probe_fused.py) with the previous
# round's library (sassy_amd/lib/libsassy_hip_r3.so, built from the round-3 commit) and with this tree's, alternating
cd "${GRAFT_REPO_ROOT:-.}"
N=${1:-3}; shift || true
for i in $(seq 1 $N); do
  for lib in libsassy_hip_r3.so libsassy_hip.so; do
    echo "$lib $(env "$@" SASSY_HIP_LIBRARY=$PWD/sassy_amd/lib/$lib python tools/probe_fused.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('lone_ms','kernel_ms','matches','fused','filtered','chunks')})")"
  done
done
