cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -s -k "config5_shape" 2>&1 | tail -5
timeout 600 python tools/bench_configs.py --configs 5 2>&1 | tail -2
