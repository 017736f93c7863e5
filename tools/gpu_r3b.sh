cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
python tools/probe_shapes.py 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3b/bench_h.json 2> gpurun_out/r3b/bench_h.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3b/bench_h.json'))
print({k:d[k] for k in ['value','ms_per_step','single_search_latency_ms','single_search_roofline_frac','dominant_kernel_ms']})
print(d['roofline']['frac'], d['roofline_search'])
print(d.get('other_configs'))
PY
timeout 900 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['value','ms_per_step','single_search_latency_ms','dominant_kernel_ms']})"
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not forced and not config4 and not beyond" > gpurun_out/r3b/tests.log 2>&1
tail -3 gpurun_out/r3b/tests.log
