cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
SASSY_HIP_DEBUG_FUSED=1 timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "fused_filter_equals" > gpurun_out/r3b/tests.log 2>&1
grep "fused launch" gpurun_out/r3b/tests.log | sort | uniq -c | head; tail -5 gpurun_out/r3b/tests.log
