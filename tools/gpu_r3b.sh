cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "qgram or count or config_3 or config3 or fuzz_small or low_complexity or rc_strand or beyond_4gib or texts_that" 2>&1 | tail -3
timeout 300 python tests/fuzz_gpu.py --seconds 150 --seed 314 --focus count 2>&1 | tail -2
SASSY_HIP_FILTER_KIND=4 timeout 300 python tests/fuzz_gpu.py --seconds 100 --seed 315 2>&1 | tail -2
