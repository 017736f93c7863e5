#!/bin/bash
# tools/collect_profiles.sh -- HERE, after `gpurun -- bash tools/refresh_profiles.sh r01`: copy the merged
# results from gpurun_out/refresh/ into the tracked profiles/ directory.
set -eu
cd "$(dirname "$0")/.."
for f in gpurun_out/refresh/*.json gpurun_out/refresh/*.csv gpurun_out/refresh/*.txt; do
  [ -s "$f" ] && cp "$f" profiles/
done
ls -la profiles/
