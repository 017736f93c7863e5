cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PROBE_PROFILE=iupac PROBE_M=200 PROBE_K=20 PROBE_C3=1
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/prof_c3b -o t -- python tools/probe_fused.py 2>/dev/null | tail -1 | cut -c1-200
python tools/timeline.py gpurun_out/prof_c3b filter_count
SASSY_HIP_TRACE_PROBE=1 python tools/probe_fused.py 2>&1 | grep "trace waves" | tail -1
