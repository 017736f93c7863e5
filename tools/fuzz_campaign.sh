#!/bin/bash
# tools/fuzz_campaign.sh <out file> [seconds per family] [seconds of the mix] -- ON THE GPU BOX: tests/fuzz_gpu.py, every case family
# on its own and then the mix, seeds of their own (the pytest slice -- tests/test_gpu_fuzz.py -- uses seed 6); one line per run.
cd "${GRAFT_REPO_ROOT:-.}"
OUT=$1; PER=${2:-150}; MIX=${3:-600}
{ echo "# tests/fuzz_gpu.py against the oracle, $(git rev-parse --short HEAD 2>/dev/null || echo tree) -- family / seed: cases, matches compared, prefilter kinds";
  seed=60
  for fam in one fused bytes_long count many encoded shard inflight reflanes ovenc; do
    seed=$((seed + 1))
    echo "--focus $fam --seed $seed --seconds $PER: $(python tests/fuzz_gpu.py --focus $fam --seed $seed --seconds $PER 2>&1 | tail -1)"
  done
  echo "mix --seed 77 --seconds $MIX: $(python tests/fuzz_gpu.py --seed 77 --seconds $MIX 2>&1 | tail -1)"; } > $OUT 2>&1
