cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
for a in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do
timeout 600 python bench.py $a --no-cpu-baseline > gpurun_out/r3b/bench_c.json 2> gpurun_out/r3b/bench_c.err
cat gpurun_out/r3b/bench_c.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['value','ms_per_step','single_search_latency_ms','single_search_roofline_frac','fused_filter_launch','dominant_kernel_ms','matches']})"
done
SASSY_HIP_WAVES_PER_CU=16 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r3b/bench_d.json 2> gpurun_out/r3b/bench_d.err
cat gpurun_out/r3b/bench_d.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['value','ms_per_step','single_search_latency_ms','single_search_roofline_frac','fused_filter_launch','dominant_kernel_ms','matches']})"
for f in 2 4; do
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --in-flight $f > gpurun_out/r3b/bench_e.json 2> gpurun_out/r3b/bench_e.err
cat gpurun_out/r3b/bench_e.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('in flight $f', {k:d[k] for k in ['value','ms_per_step','single_search_latency_ms']})"
done
