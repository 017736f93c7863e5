cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "not forced_kernel_paths and not config4" 2>&1 | tail -3
PROBE_PROFILE=iupac PROBE_M=200 PROBE_K=20 SASSY_HIP_TRACE_PROBE=1 python tools/probe_fused.py 2>&1 | grep "trace waves\|lone_ms" | tail -2 | cut -c1-200
SASSY_HIP_TRACE_PROBE=1 python tools/probe_fused.py 2>&1 | grep "trace waves\|lone_ms" | tail -2 | cut -c1-200
python tools/probe_fused.py 2>&1 | tail -1 | cut -c1-100
