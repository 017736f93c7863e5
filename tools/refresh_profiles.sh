#!/bin/bash
# tools/refresh_profiles.sh <round-tag> -- ON THE GPU BOX (via gpurun): every measurement profiles/ holds,
# written under gpurun_out/refresh/ (copy the files into profiles/ afterwards: tools/collect_profiles.sh).
set -u
TAG=${1:-r06}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/refresh
rm -rf $OUT
mkdir -p $OUT
# ---- the bench line (two searches in flight) and the driver's short form of it
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_steps20.json 2>> $OUT/bench.err
# ---- rocprofv3 of the same command: kernel stats + PMC passes, searches in flight / one at a time
bash tools/prof.sh $TAG > /dev/null 2>&1
cp gpurun_out/prof_$TAG/summary.txt $OUT/${TAG}_rocprof_summary.txt
cp $(ls gpurun_out/prof_$TAG/trace/*kernel_stats.csv gpurun_out/prof_$TAG/trace/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/${TAG}_kernel_stats.csv
bash tools/prof.sh ${TAG}_one --in-flight 1 > /dev/null 2>&1
cp gpurun_out/prof_${TAG}_one/summary.txt $OUT/${TAG}_rocprof_summary_one_at_a_time.txt
cp gpurun_out/prof_${TAG}_one/hbm_traffic.json $OUT/hbm_traffic.json
# ---- one search at a time: kernel durations, the gaps between them, the host's turnaround (fused launch / classic chain)
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_${TAG}_lone -o t -- python tools/probe_fused.py > $OUT/lone_fused.json 2>> $OUT/bench.err
SASSY_HIP_FUSED=0 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_${TAG}_lone_classic -o t -- python tools/probe_fused.py > $OUT/lone_classic.json 2>> $OUT/bench.err
{ echo "# a lone search of BASELINE config 2 (3 GB, |P| = 32, k = 3), tools/probe_fused.py under rocprofv3 --kernel-trace; tools/timeline.py";
  echo "## fused launch (default)"; tail -1 $OUT/lone_fused.json; python tools/timeline.py gpurun_out/prof_${TAG}_lone;
  echo "## classic chain (SASSY_HIP_FUSED=0)"; tail -1 $OUT/lone_classic.json; python tools/timeline.py gpurun_out/prof_${TAG}_lone_classic;
  echo "## where the fused launch's waves spend their time (SASSY_HIP_FUSED_PROBE=1: streaming / chunk DP, 100 MHz ticks; =2: no chunk DP at all)";
  SASSY_HIP_FUSED_PROBE=1 python tools/probe_fused.py 2>/dev/null | tail -1; SASSY_HIP_FUSED_PROBE=2 python tools/probe_fused.py 2>/dev/null | tail -1;
  echo "## traceback waves, microseconds per report and phase (SASSY_HIP_TRACE_PROBE=1)"; SASSY_HIP_TRACE_PROBE=1 python tools/probe_fused.py 2>&1 | grep "trace waves" | tail -2; } > $OUT/${TAG}_lone_search_timeline.txt 2>&1
# ---- the streaming DP (prefilter off): with and without the row cut-off
SASSY_HIP_PREFILTER=0 bash tools/prof.sh ${TAG}_scan --in-flight 1 > /dev/null 2>&1
cp gpurun_out/prof_${TAG}_scan/summary.txt $OUT/${TAG}_scan_kernel_rocprof_summary.txt
SASSY_HIP_PREFILTER=0 SASSY_HIP_ROW_CUT=0 bash tools/prof.sh ${TAG}_scan_nocut --in-flight 1 > /dev/null 2>&1
cp gpurun_out/prof_${TAG}_scan_nocut/summary.txt $OUT/${TAG}_scan_kernel_all_rows_rocprof_summary.txt
{ echo "# streaming DP (SASSY_HIP_PREFILTER=0), BASELINE config 2 and config 3 shapes, row cut-off on / off (SASSY_HIP_ROW_CUT=0)";
  echo "# rows the wave-voted cut-off computes per block: $(for mk in '32 3' '200 20'; do set -- $mk; SASSY_HIP_PREFILTER=0 PROBE_COUNTERS=1 PROBE_M=$1 PROBE_K=$2 python tools/probe_fused.py 2>/dev/null | tail -1 | grep -o '"rows_per_block": [0-9.]*' | sed "s/^/m=$1 k=$2 /"; done | tr '\n' ' ')";
  for cut in 1 0; do for shape in "--profile dna --pattern-len 32 --k 3" "--profile iupac --pattern-len 200 --k 20" "--profile dna --pattern-len 64 --k 6"; do
    echo "ROW_CUT=$cut $shape"; SASSY_HIP_PREFILTER=0 SASSY_HIP_ROW_CUT=$cut python bench.py $shape --steps 30 --warmup 5 --in-flight 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   ms_per_search', d['ms_per_step'], 'scan_kernel_ms', d['dominant_kernel_ms'], 'roofline_frac', d['roofline']['frac'], 'matches', d['matches'])"; done; done; } > $OUT/${TAG}_streaming_dp.txt 2>&1
# ---- searches in flight: depth, filter occupancy, chaining
{ echo "# python bench.py --steps 400 --warmup 50 (3 GB, config 2): ms per search by searches in flight and switches";
  for v in "--in-flight 1" "--in-flight 2" "--in-flight 3" "--in-flight 4"; do echo "$v: $(python bench.py --steps 400 --warmup 50 --no-cpu-baseline $v 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; done;
  for e in "SASSY_HIP_FILTER_LINEAR=8192" "SASSY_HIP_PIPE_DEPTH=3"; do echo "--in-flight 2 $e: $(env $e python bench.py --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; done; } > $OUT/${TAG}_in_flight.txt 2>&1
# ---- lane-chunk geometry: default vs the opt-in tuner, lone searches and searches in flight, several text sizes
{ echo "# bench.py --steps 300 --warmup 60 --tune-searches 40: ms per search (in flight 2) | latency of a lone search; SASSY_HIP_TUNE=1 = opt-in tuner";
  for n in 1000000000 2000000000 2700000000 3000000000 3700000000 5000000000; do for t in 0 1; do
    echo "text_bytes $n TUNE=$t: $(SASSY_HIP_TUNE=$t python bench.py --steps 300 --warmup 60 --tune-searches 40 --text-bytes $n --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"single_search_latency_ms": [0-9.]*' | tr '\n' ' ')"; done; done; } > $OUT/${TAG}_geometry_sweep.txt 2>&1
# ---- the counting filter (Iupac, long patterns): waves per workgroup around one table copy; the config-3 shape as a lone search
{ echo "# filter_count_kernel on 3 GB, tools/probe_fused.py (lone search, kernel_ms = the filter): four / sixteen waves per workgroup";
  echo "# (one-off builds, round 3: half the look-ups 0.578-0.592 ms, no look-ups 0.537 ms, no look-ups + Iupac text check 0.562 ms; WPG=4: 0.593-0.600)";
  for shape in "iupac 200 20" "iupac 32 3" "dna 100 10"; do set -- $shape; for w in 4 16; do
    echo "profile $1 m $2 k $3 SASSY_HIP_COUNT_WPG=$w: $(PROBE_PROFILE=$1 PROBE_M=$2 PROBE_K=$3 SASSY_HIP_FILTER_KIND=4 SASSY_HIP_COUNT_WPG=$w python tools/probe_fused.py 2>/dev/null | tail -1 | cut -c1-200)"; done; done;
  echo "## the config-3 shape (Iupac, m=200, k=20, N R Y W in the pattern) as a lone search: kernel timelines (tools/c3_timeline.sh)";
  echo "## default: the filter files its chunk descriptors itself, compact_chunks_kernel, list_rows_kernel (a lane per block)"; } > $OUT/${TAG}_count_filter.txt 2>&1
bash tools/c3_timeline.sh $OUT/${TAG}_count_filter.txt
bash tools/c3_timeline.sh $OUT/${TAG}_count_filter.txt SASSY_HIP_LIST_WORDS=2
bash tools/c3_timeline.sh $OUT/${TAG}_count_filter.txt SASSY_HIP_COUNT_FUSED=0
bash tools/c3_timeline.sh $OUT/${TAG}_count_filter.txt SASSY_HIP_COUNT_FUSED=0 SASSY_HIP_LIST_WORDS=2
{ echo "## its traceback waves, microseconds per report and phase"; PROBE_C3=1 PROBE_PROFILE=iupac PROBE_M=200 PROBE_K=20 SASSY_HIP_TRACE_PROBE=1 python tools/probe_fused.py 2>&1 | grep "trace waves" | tail -1; } >> $OUT/${TAG}_count_filter.txt 2>&1
# ---- does gfx950 skip masked 16-lane quarters of a VALU instruction?  (sub-wave groups for the streaming DP: no)
{ echo "# tools/ubench/exec_skip.hip: 4096 x 128 dependent v_bitop3_b32 per wave, 8 waves per SIMD, by EXEC mask";
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/exec_skip tools/ubench/exec_skip.hip 2>/dev/null && /tmp/exec_skip; } > $OUT/${TAG}_exec_mask_ubench.txt 2>&1
# ---- other configs, shapes, texts
python tools/bench_configs.py --configs 1,3,4,5 --patterns 10000 > $OUT/${TAG}_configs.json 2> $OUT/configs.err
bash tools/prof_configs.sh cfg > /dev/null 2>&1
cp gpurun_out/prof_cfg/summary.txt $OUT/${TAG}_configs_prof.txt
# config 4 at its own size: the seeded search (kernel stats + VALU / HBM counters), and the three many-pattern paths
PATTERNS=10000 CONFIGS=4 bash tools/prof_configs.sh cfg10k > /dev/null 2>&1
cp gpurun_out/prof_cfg10k/summary.txt $OUT/${TAG}_config4_seeded_prof.txt
cp $(ls gpurun_out/prof_cfg10k/trace/*kernel_stats.csv gpurun_out/prof_cfg10k/trace/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/${TAG}_config4_seeded_kernel_stats.csv
python tools/bench_encoded.py > $OUT/${TAG}_encoded_paths.txt 2> $OUT/encoded.err
{ echo "# tools/pmc_cmd.sh seed_search_kernel: counters per dispatch (sums over the waves; SQ cycle counters in units of 4 cycles), config 4 and the CRISPR guide set";
  echo "## python tools/bench_configs.py --configs 4 --patterns 10000 --steps 2"; bash tools/pmc_cmd.sh seed_search_kernel python tools/bench_configs.py --configs 4 --patterns 10000 --steps 2;
  echo "## python tools/bench_crispr.py (forward and both strands: the mean of the two)"; bash tools/pmc_cmd.sh seed_search_kernel python tools/bench_crispr.py; } > $OUT/${TAG}_seeded_pmc.txt 2>&1
{ echo "# tools/timeline_all.py: the last call of tools/bench_crispr.py --genome-like (both strands) and of tools/bench_reads.py --reads 330000: kernels and copies longer than 0.3 ms";
  for what in "bench_crispr.py --genome-like" "bench_reads.py --reads 330000"; do
    ( cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace -f csv -d $OLDPWD/gpurun_out/tl_$$ -o t -- python $OLDPWD/tools/$what > /dev/null 2>&1 )
    echo "## $what"; python tools/timeline_all.py gpurun_out/tl_$$ --gap-ms 15 | awk '{d=$5+0; if (NR==1 || d > 300) print}' | cut -c1-150; rm -rf gpurun_out/tl_$$; done; } > $OUT/${TAG}_many_pattern_timelines.txt 2>&1
python tools/bench_crispr.py > $OUT/${TAG}_crispr.json 2> $OUT/crispr.err
python tools/bench_crispr.py --genome-like >> $OUT/${TAG}_crispr.json 2>> $OUT/crispr.err
python tools/bench_crispr.py --overhang 0.5 > $OUT/${TAG}_overhang_one_text.json 2>> $OUT/crispr.err
python tools/bench_texts.py > $OUT/${TAG}_texts.json 2> $OUT/texts.err
# ---- kernel timelines of lone searches on the texts that are not i.i.d. (dense plants, N runs under Iupac, poly-A, microsatellite)
{ echo "# lone searches of tools/bench_texts.py cases under rocprofv3 --kernel-trace (tools/prof_texts.sh): the last dispatches in time order, then totals";
  TAIL_N=22 bash tools/prof_texts.sh iid_plant_4KiB repeatsN_iupac_random32 repeats_polyA repeats_family32 repeats_ACx16; } > $OUT/${TAG}_texts_timelines.txt 2>&1
# ---- this tree against the previous round's library on the same box (sassy_amd/lib/libsassy_hip_r3.so, when it was built)
if [ -f sassy_amd/lib/libsassy_hip_r3.so ]; then
  { echo "# tools/ab.sh: lone config-2 search (tools/probe_fused.py), round-3 library / this tree, alternating on one box";
    echo "## Dna"; bash tools/ab.sh 3; echo "## Iupac searcher, plain pattern (CHECK launch)"; bash tools/ab.sh 3 PROBE_PROFILE=iupac; } > $OUT/${TAG}_ab_vs_r03.txt 2>&1
fi
# ---- the multi-device searcher as the bench line's driver (one device here), and the RCCL preflight's script on gloo
python bench.py --mode inproc --gpus 1 --no-cpu-baseline > $OUT/${TAG}_bench_inproc.json 2>> $OUT/bench.err
python tools/preflight_multigpu.py --gpus 2 --backend gloo 2>/dev/null | tail -1 > $OUT/${TAG}_preflight_gloo.json
{ python tools/bench_reads.py; python tools/bench_reads.py --reads 330000; python tools/bench_reads.py --reads 330000 --fwd; python tools/bench_reads.py --overhang 0.5; python tools/bench_reads.py --reads 330000 --overhang 0.5; SASSY_HIP_OVERHANG_SEEDED=0 python tools/bench_reads.py --reads 330000 --overhang 0.5; SASSY_HIP_OVERHANG_TILED=0 python tools/bench_reads.py --overhang 0.5; } > $OUT/${TAG}_reads.json 2> $OUT/reads.err
{ python tools/probe_count.py; python tools/probe_rc.py; PROBE_M=23 PROBE_K=3 python tools/probe_rc.py; } > $OUT/${TAG}_shapes.json 2> $OUT/shapes.err
{ echo "# the paired filter (round 5) against the paths of round 4 (SASSY_HIP_PAIR=0): lone searches, 3 GB, tools/probe_short_pieces.py";
  export PROBE_SHAPES="dna:32:3,dna:23:3,iupac:23:3,dna:32:4,dna:32:5,iupac:32:5,dna:24:3,dna:27:3,dna:20:2,dna:12:1,dna:40:6,dna:48:7";
  echo "## default"; PROBE_DEFAULT_ONLY=1 python tools/probe_short_pieces.py; echo "## SASSY_HIP_PAIR=0"; SASSY_HIP_PAIR=0 PROBE_DEFAULT_ONLY=1 python tools/probe_short_pieces.py;
  echo "## counters of the fused launch, m = 23, k = 3 (tools/pmc_kernel.sh) and where its waves spend their time (SASSY_HIP_FUSED_PROBE=1 / 2: no chunk DP)";
  bash tools/pmc_kernel.sh filter_dna_kernel PROBE_M=23 PROBE_K=3; SASSY_HIP_FUSED_PROBE=1 PROBE_M=23 PROBE_K=3 python tools/probe_fused.py 2>/dev/null | tail -1; SASSY_HIP_FUSED_PROBE=2 PROBE_M=23 PROBE_K=3 python tools/probe_fused.py 2>/dev/null | tail -1;
  unset PROBE_SHAPES; } > $OUT/${TAG}_paired_filter.txt 2> $OUT/pair.err
{ echo "# tools/probe_short_pieces.py: shapes whose pigeonhole pieces are 5 or 6 rows -- default path against no prefilter (streaming DP), lone searches, 3 GB";
  python tools/probe_short_pieces.py; } > $OUT/${TAG}_short_pieces.txt 2> $OUT/short.err
{ echo "# tools/pmc_kernel.sh list_rows_kernel: config 3's chunk DP (a lane per block), counters per dispatch (sums over the waves; cycles in units of 4)";
  bash tools/pmc_kernel.sh list_rows_kernel PROBE_C3=1 PROBE_PROFILE=iupac PROBE_M=200 PROBE_K=20; } > $OUT/${TAG}_list_rows_pmc.txt 2>&1
python tools/bench_cli.py > $OUT/${TAG}_bench_cli.json 2> $OUT/bench_cli.err
python tools/cpu_probe.py > $OUT/${TAG}_host_cpus.txt 2>&1
tail -2 $OUT/*.err
ls -la $OUT
