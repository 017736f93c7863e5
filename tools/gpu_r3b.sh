cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for f in 2 3; do
for pad in default 0 8192 12288 16384 24576; do
  if [ $pad = default ]; then e=""; else e="SASSY_HIP_FILTER_LDS_PAD=$pad"; fi
  echo "rep $rep in-flight $f pad $pad: $(env $e python bench.py --steps 300 --warmup 50 --no-cpu-baseline --in-flight $f 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done; done; done
