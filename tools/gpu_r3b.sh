cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ascii or long_pattern or fuzz_small or many_pieces" > gpurun_out/r3b/tests.log 2>&1
tail -30 gpurun_out/r3b/tests.log
