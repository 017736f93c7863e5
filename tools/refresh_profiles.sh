#!/bin/bash
# tools/refresh_profiles.sh <round-tag> -- ON THE GPU BOX (via gpurun): every measurement profiles/ holds,
# written under gpurun_out/refresh/ (copy the files into profiles/ afterwards: tools/collect_profiles.sh).
set -u
TAG=${1:-r01}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/refresh
mkdir -p $OUT
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
bash tools/prof.sh $TAG > /dev/null 2>&1
cp gpurun_out/prof_$TAG/summary.txt $OUT/${TAG}_rocprof_summary.txt
cp gpurun_out/prof_$TAG/hbm_traffic.json $OUT/hbm_traffic.json
cp $(ls gpurun_out/prof_$TAG/trace/*kernel_stats.csv gpurun_out/prof_$TAG/trace/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/${TAG}_kernel_stats.csv
python tools/bench_configs.py --configs 1,3,4 --patterns 10000 > $OUT/${TAG}_configs.json 2> $OUT/configs.err
bash tools/prof_configs.sh cfg > /dev/null 2>&1
cp gpurun_out/prof_cfg/summary.txt $OUT/${TAG}_configs_prof.txt
{ python tools/bench_reads.py; python tools/bench_reads.py --reads 330000; python tools/bench_reads.py --overhang 0.5; } > $OUT/${TAG}_reads.json 2> $OUT/reads.err
{ python tools/probe_count.py; python tools/probe_rc.py; } > $OUT/${TAG}_shapes.json 2> $OUT/shapes.err
mkdir -p tools/ubench/bin
[ -x tools/ubench/bin/unaligned_read ] || hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/unaligned_read tools/ubench/unaligned_read.hip 2> /dev/null
./tools/ubench/bin/unaligned_read > $OUT/${TAG}_unaligned_read.txt 2>&1
tail -2 $OUT/*.err
ls -la $OUT
