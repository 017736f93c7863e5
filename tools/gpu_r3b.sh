cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "encoded or config4 or search_many or regressions or texts_that" 2>&1 | tail -3
timeout 300 python tests/fuzz_gpu.py --seconds 120 --seed 313 --focus encoded 2>&1 | tail -2
