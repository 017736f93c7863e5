cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
for f in "" fused count encoded many inflight bytes_long reflanes; do
  timeout 200 python tests/fuzz_gpu.py --seconds 100 --seed 4$RANDOM ${f:+--focus $f} 2>&1 | tail -1
done
