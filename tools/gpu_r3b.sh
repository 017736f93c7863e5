cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b gpurun_out/r3c
timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not forced" > gpurun_out/r3b/tests.log 2>&1
tail -5 gpurun_out/r3b/tests.log
