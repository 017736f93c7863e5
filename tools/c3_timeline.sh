#!/bin/bash
# tools/c3_timeline.sh <out file> [ENV=VAL ...] -- the config-3 shape (Iupac, m = 200, k = 20, N R Y W in the pattern) as a lone
# search under rocprofv3 --kernel-trace: per-kernel timeline (tools/timeline.py) for the given switches.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=$1; shift
D=gpurun_out/c3_$$; rm -rf $D; mkdir -p $D
env PROBE_PROFILE=iupac PROBE_M=200 PROBE_K=20 PROBE_C3=1 "$@" rocprofv3 --kernel-trace -f csv -d $D -o t -- python tools/probe_fused.py > $D/probe.json 2> $D/err.txt
{ echo "## $*"; tail -1 $D/probe.json; python tools/timeline.py $D filter_count_kernel; } >> $OUT
rm -rf $D
