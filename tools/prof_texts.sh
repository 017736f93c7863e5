#!/bin/bash
# tools/prof_texts.sh <case> [<case> ...] -- ON THE GPU BOX: kernel timeline of lone searches of tools/bench_texts.py cases
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/texts
for c in "$@"; do
  OUT=gpurun_out/texts/$c
  rm -rf $OUT; mkdir -p $OUT
  PROBE_CASE=$c PROBE_REPS=${PROBE_REPS:-6} rocprofv3 --kernel-trace -f csv -d $OUT/trace -o t -- python tools/probe_text_case.py > $OUT/line.json 2> $OUT/err.txt
  { echo "## $c"; tail -1 $OUT/line.json; python tools/trace_tail.py $OUT/trace ${TAIL_N:-24}; } > $OUT/timeline.txt 2>&1
  cat $OUT/timeline.txt
done
