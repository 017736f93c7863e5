#!/bin/bash
# tools/collect_profiles.sh <round-tag> -- HERE, after `gpurun -- bash tools/refresh_profiles.sh <round-tag>`: copy the merged
# results from gpurun_out/refresh/ into the tracked profiles/ directory.  Only files of THIS round are written: names that
# begin with the tag, and the three untagged "latest" files (hbm_traffic.json, lone_fused.json, lone_classic.json).  A file
# that carries another round's tag is an earlier round's evidence: refused, loudly (round 4 once overwrote r02_* this way).
set -eu
TAG=${1:?usage: tools/collect_profiles.sh <round-tag, e.g. r05>}
cd "$(dirname "$0")/.."
bad=0
for f in gpurun_out/refresh/*.json gpurun_out/refresh/*.csv gpurun_out/refresh/*.txt; do
  [ -s "$f" ] || continue
  b=$(basename "$f")
  case "$b" in
    ${TAG}_*|hbm_traffic.json|lone_fused.json|lone_classic.json) cp "$f" profiles/ ;;
    r[0-9][0-9]_*) echo "REFUSED: $b belongs to another round (tag $TAG)" >&2; bad=1 ;;
    *) ;;  # scratch (*.err logs and the like)
  esac
done
ls -la profiles/ | tail -40
exit $bad
