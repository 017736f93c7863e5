cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pinned_block or fuzz_regressions or in_flight or fused" > gpurun_out/r3b/tests.log 2>&1
tail -5 gpurun_out/r3b/tests.log
