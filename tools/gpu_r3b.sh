cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3b/bench_g.json 2> gpurun_out/r3b/bench_g.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3b/bench_g.json'))
print({k:d[k] for k in ['value','ms_per_step','single_search_latency_ms','single_search_roofline_frac','dominant_kernel_ms']})
print(d.get('other_configs'))
print(d.get('cpu_baseline'))
print(d.get('h2d_inclusive'))
PY
