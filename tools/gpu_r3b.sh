cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "merge_shards or multi_searcher or stays_on_its_device or two_ranks or bench_launches" > gpurun_out/r3b/tests.log 2>&1
tail -30 gpurun_out/r3b/tests.log
