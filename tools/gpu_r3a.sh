set -x
mkdir -p gpurun_out/r3a
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or fuzz_regressions or fuzz_small or low_complexity or config1 or full_size_3gb or device_resident or shard_seam or in_flight or texts_that_are_not_iid" > gpurun_out/r3a/tests.log 2>&1
tail -15 gpurun_out/r3a/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
cat gpurun_out/r3a/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['value','ms_per_step','single_search_latency_ms','single_search_latency_with_kernel_events_ms','single_search_roofline_frac','fused_filter_launch','dominant_kernel_ms','matches']}); print(d['roofline']); print(d.get('other_configs'))"
SASSY_HIP_FUSED=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3a/bench_unfused.json 2> gpurun_out/r3a/bench_unfused.err
cat gpurun_out/r3a/bench_unfused.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['value','ms_per_step','single_search_latency_ms','single_search_latency_with_kernel_events_ms','fused_filter_launch','dominant_kernel_ms','matches']})"
